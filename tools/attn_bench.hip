// Micro-benchmark of the long-sequence attention kernel (kirag_amd/csrc/encoder.hip: k_attn_dma) on synthetic f16 Q / K / V^T of B sequences x S tokens,
// 16 heads x 64:   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DKR_STAMP_ATTN] tools/attn_bench.hip -o gpurun_out/attn_bench && ./gpurun_out/attn_bench [B S]
// Prints us per launch (HIP events), TFLOP/s of the two MFMA products, a checksum of ctx, and with -DKR_STAMP_ATTN the per-wave cycle split of the chunk loop.
#define KR_ENC_BUILD_F16 1
#include "../kirag_amd/csrc/encoder.hip"

#include <cstdio>
#include <random>

namespace kr {
std::string& last_error_ref() { static thread_local std::string e; return e; }
int fail(int code, const char*, ...) { return code; }
int select_device(int) { return 0; }
bool is_device_pointer(const void*) { return true; }
std::atomic<int> g_force_exact{0};
}
using namespace kr::enc_f16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128, S = argc > 2 ? atoi(argv[2]) : 512, H = 1024, heads = 16;
    const int64_t T = (int64_t)B * S, ldv = T + 64;
    std::vector<uint16_t> hq((size_t)T * H), hk((size_t)(T + 64) * H), hv((size_t)H * ldv);
    std::mt19937 rng(1); std::normal_distribution<float> N01(0.f, 1.f);
    auto f16 = [](float f) { _Float16 h = (_Float16)f; uint16_t u; __builtin_memcpy(&u, &h, 2); return u; };
    for (auto& v : hq) v = f16(N01(rng) * 0.18f);        // scores in log2 units with a spread of a few units
    for (auto& v : hk) v = f16(N01(rng));
    for (auto& v : hv) v = f16(N01(rng));
    std::vector<int> off(B), nk(B, S), nq(B, S);
    for (int b = 0; b < B; ++b) off[b] = b * S;
    uint16_t *q, *k, *vT, *ctx; int *d_off, *d_nk, *d_nq;
    CK(hipMalloc(&q, hq.size() * 2)); CK(hipMalloc(&k, hk.size() * 2)); CK(hipMalloc(&vT, hv.size() * 2)); CK(hipMalloc(&ctx, (size_t)T * H * 2));
    CK(hipMalloc(&d_off, B * 4)); CK(hipMalloc(&d_nk, B * 4)); CK(hipMalloc(&d_nq, B * 4));
    CK(hipMemcpy(q, hq.data(), hq.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(k, hk.data(), hk.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(vT, hv.data(), hv.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_off, off.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_nk, nk.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_nq, nq.data(), B * 4, hipMemcpyHostToDevice));
    const int nqt = (S + 31) / 32, qgroups = (nqt + ADMA_QT - 1) / ADMA_QT;
    const int extra_lds = argc > 3 ? atoi(argv[3]) : 0;      // > 32 KiB: only one block fits a CU (one wave per SIMD): how much do two co-resident blocks overlap?
    if (extra_lds) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_dma), hipFuncAttributeMaxDynamicSharedMemorySize, ADMA_LDS + extra_lds));
    auto launch = [&] {
        hipLaunchKernelGGL(k_attn_dma, dim3(((B * heads + 7) / 8) * 8 * qgroups), dim3(ADMA_THREADS), ADMA_LDS + extra_lds, 0, q, k, vT, ldv, d_off, d_nk, d_nq, H, T, ctx,
                           heads, B, qgroups);
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
#ifdef KR_STAMP_ATTN
    unsigned long long zero[8] = {}; CK(hipMemcpyToSymbol(HIP_SYMBOL(kr_attn_stamps), zero, sizeof zero));
#endif
    const int reps = 20;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, fl = 4.0 * B * heads * (double)S * S * 64;
    std::vector<uint16_t> hc((size_t)T * H);
    CK(hipMemcpy(hc.data(), ctx, hc.size() * 2, hipMemcpyDeviceToHost));
    double sum = 0; for (size_t i = 0; i < hc.size(); i += 97) { _Float16 h; __builtin_memcpy(&h, &hc[i], 2); sum += (double)(float)h; }
    printf("k_attn_dma %d x %d (+%d B LDS): %.1f us per launch, %.0f TFLOP/s (QK^T + PV), ctx checksum %.6f\n", B, S, extra_lds, us, fl / us / 1e6, sum);
#ifdef KR_STAMP_ATTN
    unsigned long long st[8]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(kr_attn_stamps), sizeof st));
    const double w = (double)st[4], chunks = (double)(S / 64);
    printf("  per wave: lifetime %.0f cycles; per chunk: barrier/wait/issue %.0f, tile 0 step %.0f, tile 1 step %.0f; outside the chunk loop %.0f\n", st[3] / w,
           st[0] / w / chunks, st[1] / w / chunks, st[2] / w / chunks, (st[3] - st[0] - st[1] - st[2]) / w);
#endif
    return 0;
}
