// Shared device/host helpers for libkirag_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <cstdio>
#include <cstdarg>
#include <atomic>

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
bool is_device_pointer(const void* p);   // hipMalloc memory (as opposed to pageable / pinned host memory)
extern std::atomic<int> g_force_exact;   // kr_set_option("force_exact_scores")
extern std::atomic<int> g_byte_min_rows; // kr_set_option("debug_byte_min_rows"): index size from which they do (default 2^19; tests lower it)
extern std::atomic<int> g_eps8_permille; // kr_set_option("debug_eps8_permille"): MUTATION hook of the tests - the byte pre-scan's error bound eps8 is multiplied by value / 1000 (1000 = the bound)
extern std::atomic<int> g_byte_prescan;  // kr_set_option("byte_prescan"): small query blocks may take the int8 pre-scan (search.hip, byte_final_round)
extern std::atomic<unsigned long long> g_vmm_min_reserve;   // kr_set_option("debug_vmm_min_reserve_mib"): smallest address range reserved for a large index (default 16 GiB)
extern std::atomic<unsigned long long> g_va_retired_bias;   // kr_set_option("debug_va_retired_tib"): test hook of the index's address-space budget

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
    static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
struct F16 {
    static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
    static __device__ __forceinline__ float to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
    static __device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
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

// ---- canonical score ----------------------------------------------------------------------------------------------------------------------
// score(q, x) = RN32( EXACT inner product of the fp32 inputs ): the exact real sum rounded once to fp32, round-to-nearest-even.  A mathematical
// definition, independent of any summation order; oracle/search_c.c computes it sequentially (plain fp64 + certified rounding, integer
// super-accumulator), tests/golden/g9_exact_dot.npz holds known answers from Python rationals.  Here:
//   fast path   lane l sums its share of the (exact) fp64 products p_i and of |p_i|, two XOR butterflies give S and A on every lane; for ANY
//               summation order |S - exact| <= (d-1) u A / (1 - (d-1) u), u = 2^-53, so the exact sum lies in [S - E, S + E] with
//               E = (d + 4) * 1.2e-16 * A; if both ends round to the same float, that float is the canonical score.
//   exact path  (about 2 in 10^6 scores of unit vectors; ALWAYS with force_exact, a test hook) the wave adds every product as an integer into a
//               704-bit fixed-point accumulator in LDS (ds_add_u64 on 22 limbs of 32 payload bits, bit 0 = 2^-298), lane 0 propagates the
//               carries and rounds the exact integer once.
constexpr int EXACT_NLIMB = 22;

__device__ __forceinline__ void f32_decode(float f, uint32_t& sign, uint64_t& m, int& e) {   // |f| = m * 2^e, finite f
    const uint32_t u = __float_as_uint(f);
    sign = u >> 31;
    const uint32_t ex = (u >> 23) & 0xffu, fr = u & 0x7fffffu;
    if (ex == 0) { m = fr; e = -149; }
    else { m = fr | 0x800000u; e = (int)ex - 150; }
}

// wave-cooperative; q, x: d finite floats (d % 4 == 0, 16-B aligned); limbs: EXACT_NLIMB words of LDS private to this wave.  Every lane returns the score.
__device__ __forceinline__ float exact_dot_rn32_wave(const float* __restrict__ q, const float* __restrict__ x, int d, int lane, unsigned long long* limbs) {
    if (lane < EXACT_NLIMB) limbs[lane] = 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane * 4; i < d; i += 256) {
        const float4 a = *reinterpret_cast<const float4*>(q + i);
        const float4 b = *reinterpret_cast<const float4*>(x + i);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint32_t sa, sb; uint64_t ma, mb; int ea, eb;
            f32_decode(av[c], sa, ma, ea);
            f32_decode(bv[c], sb, mb, eb);
            const uint64_t M = ma * mb;                          // < 2^48, exact
            if (M == 0ull) continue;
            const int o = ea + eb + 298;                         // bit offset of M's LSB: 0 .. 506
            const int limb = o >> 5, sh = o & 31;
            unsigned long long p0 = (M << sh) & 0xffffffffull;
            unsigned long long p1 = (M >> (32 - sh)) & 0xffffffffull;
            unsigned long long p2 = sh > 16 ? (M >> (64 - sh)) : 0ull;
            if (sa ^ sb) { p0 = 0ull - p0; p1 = 0ull - p1; p2 = 0ull - p2; }   // two's complement: the limbs are signed
            atomicAdd(&limbs[limb], p0);
            if (p1) atomicAdd(&limbs[limb + 1], p1);
            if (p2) atomicAdd(&limbs[limb + 2], p2);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        long long carry = 0;
        for (int i = 0; i < EXACT_NLIMB; ++i) {                  // digits in [0, 2^32); the carry out of the top limb is 0 (value >= 0) or -1
            const long long t = (long long)limbs[i] + carry;
            carry = t >> 32;
            limbs[i] = (unsigned long long)(t - carry * 4294967296LL);
        }
        const bool neg = carry < 0;
        if (neg) {                                               // magnitude = 2^(32 NLIMB) - digits
            unsigned long long c = 1ull;
            for (int i = 0; i < EXACT_NLIMB; ++i) { const unsigned long long v = ((~limbs[i]) & 0xffffffffull) + c; limbs[i] = v & 0xffffffffull; c = v >> 32; }
        }
        int top = EXACT_NLIMB - 1;
        while (top >= 0 && limbs[top] == 0ull) --top;
        float r = 0.0f;                                          // exact zero: +0
        if (top >= 0) {
            const int hb = 31 - __clz((unsigned int)limbs[top]);
            const int P = top * 32 + hb;                         // leading bit; value = magnitude * 2^-298
            const int lsb = P - 23 > 149 ? P - 23 : 149;         // fp32 quantum: 24 significant bits, never below 2^-149
            const int a0 = lsb >> 5, s0 = lsb & 31;
            const unsigned long long win = limbs[a0] | ((a0 + 1 < EXACT_NLIMB ? limbs[a0 + 1] : 0ull) << 32);
            unsigned long long mant = (win >> s0) & ((1ull << (P - lsb + 1)) - 1ull);
            const int rb = lsb - 1;                              // >= 148
            const unsigned long long rl = limbs[rb >> 5];
            const bool rnd = (rl >> (rb & 31)) & 1ull;
            bool sticky = (rl & ((1ull << (rb & 31)) - 1ull)) != 0ull;
            for (int i = (rb >> 5) - 1; i >= 0 && !sticky; --i) sticky = limbs[i] != 0ull;
            if (rnd && (sticky || (mant & 1ull))) ++mant;
            r = ldexpf((float)mant, lsb - 298);                  // exact: mant <= 2^24; overflow -> inf; mant == 0 -> 0
            if (neg) r = -r;
        }
        limbs[0] = (unsigned long long)__float_as_uint(r);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const float out = __uint_as_float((unsigned int)limbs[0]);
    __builtin_amdgcn_wave_barrier();
    return out;
}

// S, A (wave-uniform: fp64 sum of the products and of their magnitudes, any order) -> canonical score
__device__ __forceinline__ float canonical_finish(double S, double A, const float* __restrict__ q, const float* __restrict__ x, int d, int lane,
                                                  unsigned long long* limbs, bool force_exact) {
    if (!(A <= 1.7e308)) return (float)S;                        // inf / NaN inputs propagate as in IEEE arithmetic
    const double E = (double)(d + 4) * 1.2e-16 * A;
    const float lo = (float)(S - E), hi = (float)(S + E);
    if (!force_exact && __float_as_uint(lo) == __float_as_uint(hi)) return lo;
    return exact_dot_rn32_wave(q, x, d, lane, limbs);
}

// canonical scores of TWO rows at once with the query chunks already in registers (qr[j] = q[lane*4 + j*256 .. +3], zero beyond d): every global
// load of both rows is issued before the first use, so a wave pays one memory round trip per pair of rows instead of one per 256-element step
// (the re-rank gathers one 4-KiB fp32 row per candidate from HBM).  NCH = number of 256-element steps (d <= 256 NCH).  q points at the same
// query in memory (exact path only).
template <int NCH> struct RowPair { float4 a[NCH], b[NCH]; };   // two fp32 rows, lane-sliced like the query registers

template <int NCH>
__device__ __forceinline__ void load_rows2(RowPair<NCH>& r, const float* __restrict__ x0, const float* __restrict__ x1, int d, int lane) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int i = lane * 4 + j * 256;
        r.a[j] = make_float4(0.f, 0.f, 0.f, 0.f); r.b[j] = r.a[j];
        if (i < d) { r.a[j] = *reinterpret_cast<const float4*>(x0 + i); r.b[j] = *reinterpret_cast<const float4*>(x1 + i); }
    }
}

// The products are exact in fp64 (24 x 24 bits), so s + q x rounds once either way: fma(q, x, s) is bit-identical to the product-then-add form and one
// instruction less per element; |q x| = |q| |x| takes the absolute values as free source modifiers.
template <int NCH>
__device__ __forceinline__ void score_rows2(const float4 (&qr)[NCH], const float* __restrict__ q, const RowPair<NCH>& r, const float* __restrict__ x0,
                                            const float* __restrict__ x1, int d, int lane, unsigned long long* limbs, bool force_exact, float& f0, float& f1) {
    double s0 = 0.0, s1 = 0.0, m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (lane * 4 + j * 256 < d) {
            const float qv[4] = {qr[j].x, qr[j].y, qr[j].z, qr[j].w};
            const float av[4] = {r.a[j].x, r.a[j].y, r.a[j].z, r.a[j].w}, bv[4] = {r.b[j].x, r.b[j].y, r.b[j].z, r.b[j].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const double qd = (double)qv[c], ad = (double)av[c], bd = (double)bv[c];
                s0 = __builtin_fma(qd, ad, s0); m0 = __builtin_fma(__builtin_fabs(qd), __builtin_fabs(ad), m0);
                s1 = __builtin_fma(qd, bd, s1); m1 = __builtin_fma(__builtin_fabs(qd), __builtin_fabs(bd), m1);
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s0 += shfl_xor_f64(s0, m); s1 += shfl_xor_f64(s1, m);
        m0 += shfl_xor_f64(m0, m); m1 += shfl_xor_f64(m1, m);
    }
    f0 = canonical_finish(s0, m0, q, x0, d, lane, limbs, force_exact);
    f1 = canonical_finish(s1, m1, q, x1, d, lane, limbs, force_exact);
}

template <int NCH>
__device__ __forceinline__ void canonical_score_wave2(const float4 (&qr)[NCH], const float* __restrict__ q, const float* __restrict__ x0,
                                                      const float* __restrict__ x1, int d, int lane, unsigned long long* limbs, bool force_exact,
                                                      float& f0, float& f1) {
    RowPair<NCH> r;
    load_rows2<NCH>(r, x0, x1, d, lane);
    score_rows2<NCH>(qr, q, r, x0, x1, d, lane, limbs, force_exact, f0, f1);
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
