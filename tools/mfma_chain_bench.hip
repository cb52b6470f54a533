// Dependent-accumulate latency of v_mfma_f32_32x32x16_f16 on gfx950: NCH independent accumulator chains issued round-robin by ONE wave per SIMD
// (and by two), cycles per MFMA from s_memtime around a long unrolled loop.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_chain_bench.hip -o tools/bin/mfma_chain_bench && tools/bin/mfma_chain_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int NCH>
__global__ __launch_bounds__(256) void k_chain(const f16x8* in, float* out, unsigned long long* cyc, int iters) {
    f16x8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = (float)c;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) s += acc[c][0] + acc[c][15];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (s == 1234.5f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NCH>
void run(const f16x8* in, float* out, unsigned long long* cyc, int blocks_per_cu) {
    const int iters = 2000;
    hipLaunchKernelGGL(k_chain<NCH>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, in, out, cyc, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_chain<NCH>, dim3(256 * blocks_per_cu), dim3(256), 0, 0, in, out, cyc, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    // s_memtime ticks at a fixed 100 MHz on this part?  report raw ticks per MFMA and let the caller calibrate against the 8-chain row
    printf("  %d chain(s), %d wave(s) per SIMD: %.2f memtime ticks per MFMA of one wave\n", NCH, blocks_per_cu, (double)h / ((double)iters * 8 * NCH));
}

int main() {
    f16x8* in; float* out; unsigned long long* cyc;
    CK(hipMalloc(&in, 512 * 16)); CK(hipMemset(in, 0, 512 * 16)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 8));
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<1>(in, out, cyc, bpc); run<2>(in, out, cyc, bpc); run<3>(in, out, cyc, bpc); run<4>(in, out, cyc, bpc); run<6>(in, out, cyc, bpc); run<8>(in, out, cyc, bpc);
    }
    return 0;
}
