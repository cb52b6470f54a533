// Shared device/host helpers for libkirag_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <cstdio>
#include <cstdarg>

#include "../../include/kirag_amd.h"

namespace kr {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- error plumbing -----------------------------------------------------------------------------------------
std::string& last_error_ref();
int fail(int code, const char* fmt, ...);

#define KR_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess)                                                                          \
            return kr::fail(_e == hipErrorOutOfMemory ? KR_ENOMEM : KR_EHIP, "%s failed: %s (%s:%d)", \
                            #expr, hipGetErrorString(_e), __FILE__, __LINE__);                         \
    } while (0)

#define KR_TRY(expr)             \
    do {                         \
        int _rc = (expr);        \
        if (_rc != 0) return _rc; \
    } while (0)

int select_device(int device);  // hipSetDevice + arch check (gfx950)

// ---- 16-bit element tags ------------------------------------------------------------------------------------
struct BF16 {
    static __device__ __forceinline__ uint16_t from_f32(float f) {
        // round-to-nearest-even, NaN stays NaN; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 (one VALU op)
        return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f));
    }
    static __device__ __forceinline__ float to_f32(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
    static __device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
struct F16 {
    static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
    static __device__ __forceinline__ float to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
    static __device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

// ---- order-preserving float <-> uint ------------------------------------------------------------------------
// larger float  <=> larger uint; NaN maps to 0 (ranks below everything, never beats a real score)
__host__ __device__ __forceinline__ uint32_t f32_ord(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __builtin_bit_cast(uint32_t, f);
#else
    __builtin_memcpy(&u, &f, 4);
#endif
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord_f32(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float f;
#if defined(__HIP_DEVICE_COMPILE__)
    f = __builtin_bit_cast(float, u);
#else
    __builtin_memcpy(&f, &u, 4);
#endif
    return f;
}
// 64-bit sort key: descending key order == (score desc, row asc)
__host__ __device__ __forceinline__ uint64_t make_key(float score, uint32_t row) {
    return ((uint64_t)f32_ord(score) << 32) | (uint64_t)(~row);
}
__host__ __device__ __forceinline__ float key_score(uint64_t k) { return ord_f32((uint32_t)(k >> 32)); }
__host__ __device__ __forceinline__ uint32_t key_row(uint64_t k) { return ~(uint32_t)k; }

__device__ __forceinline__ double shfl_xor_f64(double v, int m) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, m, 64);
    hi = __shfl_xor(hi, m, 64);
    return __hiloint2double(hi, lo);
}

// canonical inner product (bit-for-bit twin of oracle/search_c.c kr_oracle_dot):
// lane l accumulates, in increasing i, the exact products of the elements with ((i>>2)&63)==l, then a
// 6-stage XOR butterfly m = 32..1.  q and x point at d floats (d % 4 == 0, 16-B aligned rows).
__device__ __forceinline__ double canonical_dot_wave(const float* __restrict__ q, const float* __restrict__ x, int d, int lane) {
    double acc = 0.0;
    for (int i = lane * 4; i < d; i += 256) {
        const float4 a = *reinterpret_cast<const float4*>(q + i);
        const float4 b = *reinterpret_cast<const float4*>(x + i);
        acc += (double)a.x * (double)b.x;
        acc += (double)a.y * (double)b.y;
        acc += (double)a.z * (double)b.z;
        acc += (double)a.w * (double)b.w;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += shfl_xor_f64(acc, m);
    return acc;
}

// Same canonical score for TWO rows at once with the query chunks already in registers (qr[j] = q[lane*4 + j*256 .. +3], zero beyond d):
// every global load of both rows is issued before the first use, so a wave pays one memory round trip per pair of rows instead of one
// per 256-element step (the re-rank gathers one 4-KiB fp32 row per candidate from HBM).  NCH = number of 256-element steps (d <= 256 NCH).
// The per-lane summation order is exactly canonical_dot_wave's (products with zero-filled tails add +0.0, which changes nothing).
template <int NCH>
__device__ __forceinline__ void canonical_dot_wave2(const float4 (&qr)[NCH], const float* __restrict__ x0, const float* __restrict__ x1, int d, int lane,
                                                    double& e0, double& e1) {
    float4 a[NCH], b[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int i = lane * 4 + j * 256;
        a[j] = make_float4(0.f, 0.f, 0.f, 0.f); b[j] = a[j];
        if (i < d) { a[j] = *reinterpret_cast<const float4*>(x0 + i); b[j] = *reinterpret_cast<const float4*>(x1 + i); }
    }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (lane * 4 + j * 256 < d) {
            s0 += (double)qr[j].x * (double)a[j].x; s0 += (double)qr[j].y * (double)a[j].y; s0 += (double)qr[j].z * (double)a[j].z; s0 += (double)qr[j].w * (double)a[j].w;
            s1 += (double)qr[j].x * (double)b[j].x; s1 += (double)qr[j].y * (double)b[j].y; s1 += (double)qr[j].z * (double)b[j].z; s1 += (double)qr[j].w * (double)b[j].w;
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { s0 += shfl_xor_f64(s0, m); s1 += shfl_xor_f64(s1, m); }
    e0 = s0; e1 = s1;
}

// in-LDS bitonic sort of n (power of two) uint64 keys, DESCENDING, by a block of nthreads threads
__device__ __forceinline__ void bitonic_sort_desc(uint64_t* s, int n, int tid, int nthreads) {
    for (int k = 2; k <= n; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = tid; t < (n >> 1); t += nthreads) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j clear
                const int p = i | j;
                const bool desc = ((i & k) == 0);
                const uint64_t a = s[i], b = s[p];
                if ((a < b) == desc) { s[i] = b; s[p] = a; }
            }
        }
    }
    __syncthreads();
}

static inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

}  // namespace kr
