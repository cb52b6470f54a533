// Stand-alone check of the residual stream's byte codec (kirag_amd/csrc/encoder.hip: lo_encode / lo_decode, f16 build): random values through the
// device functions, compared with hi alone.  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/lo_codec_check.hip -o tools/bin/lo_codec_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <random>
#include <vector>
constexpr int LO_SH = 5;
__device__ __forceinline__ unsigned int lo_encode(float o, float hf) {
    const int d = (int)(__builtin_bit_cast(unsigned int, o) - __builtin_bit_cast(unsigned int, hf));
    const int t = (d + ((1 << (LO_SH - 1)) + (128 << LO_SH))) >> LO_SH;
    return (unsigned int)min(max(t, 0), 255);
}
__device__ __forceinline__ float lo_decode(unsigned int byte, float hf) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned int, hf) + (byte << LO_SH) - (128u << LO_SH));
}
__global__ void k(const float* x, float* dec, float* hi, unsigned char* lo, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float o = x[i];
    const float hf = (float)(_Float16)o;
    const unsigned int b = lo_encode(o, hf);
    lo[i] = (unsigned char)b; hi[i] = hf; dec[i] = lo_decode(b, hf);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n);
    std::mt19937 g(1); std::normal_distribution<float> N(0.f, 1.f); std::uniform_real_distribution<float> U(-6.f, 3.f);
    for (auto& v : x) v = N(g) * std::exp(U(g));
    float *dx, *dd, *dh; unsigned char* dl;
    hipMalloc(&dx, n * 4); hipMalloc(&dd, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&dl, n);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dd, dh, dl, n);
    std::vector<float> dec(n), hi(n);
    hipMemcpy(dec.data(), dd, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hi.data(), dh, n * 4, hipMemcpyDeviceToHost);
    double eh = 0, ed = 0; long worse = 0;
    for (int i = 0; i < n; ++i) { const double a = std::fabs((double)hi[i] - x[i]), b = std::fabs((double)dec[i] - x[i]); eh += a / std::fabs(x[i]); ed += b / std::fabs(x[i]); worse += b > a; }
    printf("mean relative error: hi alone %.3e, hi + lo %.3e; decoded worse than hi alone: %ld of %d\n", eh / n, ed / n, worse, n);
    return 0;
}
