// Micro-benchmark of the projection epilogue's global-store pattern (tools/store_bench.hip; hipcc -O3 --offload-arch=gfx950).
// 256 blocks x 512 threads, each block writes `tiles` 256 x 256 bf16 output tiles of a [T, F] row-major matrix (row pitch F * 2 bytes):
//   pattern 0  as the epilogue does: per store instruction 8 rows x 128 B (wave w owns 128 rows x 64 columns)
//   pattern 1  per store instruction 2 rows x 512 B (wave w owns 32 rows x 256 columns)
// each with plain or non-temporal 16-byte stores.  Prints GB/s of the burst.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int PATTERN, bool NT>
__global__ __launch_bounds__(512) void k_store(unsigned short* out, int64_t ld, int tiles_m, int tiles_n, int tiles_per_block) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const u32x4 v = {threadIdx.x, blockIdx.x, 0x3f803f80u, 0x3f803f80u};
    for (int t = 0; t < tiles_per_block; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        if (tile >= tiles_m * tiles_n) return;
        const int tm = tile / tiles_n, tn = tile % tiles_n;
        unsigned short* base = out + (int64_t)tm * 256 * ld + tn * 256;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int row, col;
            if (PATTERN == 0) { row = (w >> 2) * 128 + (i >> 2) * 32 + (i & 3) * 8 + (lane >> 3); col = (w & 3) * 64 + (lane & 7) * 8; }
            else { row = w * 32 + i * 2 + (lane >> 5); col = (lane & 31) * 8; }
            u32x4* dst = reinterpret_cast<u32x4*>(base + (int64_t)row * ld + col);
            if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
    }
}

int main() {
    const int F = 4096; const int64_t T = 131072;
    unsigned short* out; CK(hipMalloc(&out, (size_t)T * F * 2));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int tpb : {1, 4, 32}) {
        const int tiles_n = F / 256, tiles_m = (256 * tpb + tiles_n - 1) / tiles_n;
        auto run = [&](auto kern, const char* name) {
            for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, (int64_t)F, tiles_m, tiles_n, tpb);
            CK(hipDeviceSynchronize());
            float best = 1e9f;
            for (int it = 0; it < 10; ++it) {
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, (int64_t)F, tiles_m, tiles_n, tpb); CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double bytes = 256.0 * tpb * 256 * 256 * 2;
            printf("tiles/block %2d  %-28s %8.1f us  %7.1f GB/s\n", tpb, name, best * 1e3, bytes / best / 1e6);
        };
        run(k_store<0, false>, "8 rows x 128 B, plain");
        run(k_store<0, true>, "8 rows x 128 B, nt");
        run(k_store<1, false>, "2 rows x 512 B, plain");
        run(k_store<1, true>, "2 rows x 512 B, nt");
    }
    // per-CU store rate: fewer blocks (one per CU at most), each writing 32 tiles — is 128 KiB per tile bound by the CU's own store path or by the chip's?
    for (int blocks : {8, 32, 64, 128, 256}) {
        const int tpb = 32;
        const int tiles_n = F / 256, tiles_m = (blocks * tpb + tiles_n - 1) / tiles_n;
        for (int nt = 0; nt < 2; ++nt) {
            float best = 1e9f;
            for (int it = 0; it < 8; ++it) {
                CK(hipEventRecord(e0));
                if (nt) hipLaunchKernelGGL((k_store<0, true>), dim3(blocks), dim3(512), 0, 0, out, (int64_t)F, tiles_m, tiles_n, tpb);
                else hipLaunchKernelGGL((k_store<0, false>), dim3(blocks), dim3(512), 0, 0, out, (int64_t)F, tiles_m, tiles_n, tpb);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double bytes = (double)blocks * tpb * 256 * 256 * 2;
            printf("blocks %3d  %-6s %8.1f us  %7.1f GB/s total  %6.1f GB/s per block  (128 KiB tile in %.2f us)\n", blocks, nt ? "nt" : "plain", best * 1e3, bytes / best / 1e6,
                   bytes / best / 1e6 / blocks, best * 1e3 / tpb);
        }
    }
    return 0;
}
