// Seam microbenchmark, second version: a TIMING MODEL of one 32-token BERT-large forward as (a) one persistent launch with in-kernel hand-offs and
// (b) one launch per phase — the go / no-go measurement for VERDICT r04 item 1 (a single-launch small-batch forward).  Unlike tools/seam_bench.hip it
// also streams the WEIGHTS: every participant of a GEMM phase reads its 32 x K slice of a 24 x 24-MiB weight buffer (cold: HBM) through LDS, and the
// persistent form may request its NEXT phase's weights before it waits for the activations (the one thing a launch boundary cannot do).
//   phases per layer (participants, weight bytes per participant, activation bytes read, MFMA chain, bytes written per participant):
//     QKV  96  64K  64K  64  2K | attention 16  0  12K  16  4K | out-proj 32  64K  64K  64  2K | LN 8  0  20K  0  12K |
//     FF1 128  64K  64K  64  2K | FF2 32  256K 256K 256  2K   | LN 8  0  20K  0  12K
//   fusion variant (-DFUSE=1): LN folded into the consumer as a prologue (QKV / FF1 read 160K instead of 64K, the LN phases disappear): 5 hops per layer.
// Hand-off forms (persistent): R1 payload (sc1 stores, every wave drains), per-phase arrival counters SHARDED over 8 lines (blockIdx & 7), the polling lane
// reads the 8 shards with one 8-lane sc1 load per poll; SYNC=0: agent acquire after the poll, plain LDS-DMA loads; SYNC=1: no acquire, LDS-DMA loads with sc1.
// Every activation byte is checked against the epoch its producer wrote (a stale line is an error, not a slow-down).
// Build: hipcc -O3 --offload-arch=gfx950 tools/seam_bench2.hip -o tools/bin/seam_bench2 ; run: tools/bin/seam_bench2 [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

constexpr int MAXP = 7;
constexpr int THREADS = 320;
constexpr int LAYERS = 24;
constexpr int ACT_BYTES = 256 * 1024;
constexpr size_t W_LAYER = (size_t)24 << 20;
constexpr unsigned SPIN_MAX = 2000000u;
constexpr int LDS_W = 64 * 1024;           // weight ring
constexpr int LDS_A = 64 * 1024;           // activation staging (64-KiB steps: a step is one memory round trip with 64 KiB in flight)

struct Phase { int parts, wbytes, rd, nmfma, wr; size_t woff; };
struct Params {
    unsigned* cnt;       // [LAYERS * np][8 shards x 32 words]
    char* act[2];
    const char* W;
    unsigned* err;       // [0] stale, [1] timeouts, [2] checksum sink
    int np;
    Phase ph[MAXP];
    int only;            // launch-per-phase mode: the global phase this launch runs
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void store_sc1(char* base, int off, u32x4 v) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
}

// weights [w0, w0 + n) of this block's slice -> ring bytes [r0, ...) by the 4 staging waves (1-KiB pieces); no wait
__device__ __forceinline__ void stage_weights(const char* src, int w0, int n, char* ring, int wave, int lane) {
    if (wave < 1) return;
    for (int piece = wave - 1; piece * 1024 < n; piece += 4)
        __builtin_amdgcn_global_load_lds((gbl_void*)(src + w0 + piece * 1024 + lane * 16), (lds_void*)(ring + ((w0 + piece * 1024) & (LDS_W - 1))), 16, 0, 2);   // nt: read once
}

template <int SYNC, bool PERSIST>
__global__ __launch_bounds__(THREADS) void k_model(Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    char* abuf = smem + LDS_W;
    int* s_flag = reinterpret_cast<int*>(smem + LDS_W + LDS_A);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = LAYERS * p.np;
    if (tid == 0) *s_flag = 0;
    __syncthreads();
    int prefetched_g = -1, prefetched_n = 0;                 // weights of global phase prefetched_g already requested: first prefetched_n bytes
    const int g_lo = PERSIST ? 0 : p.only, g_hi = PERSIST ? total : p.only + 1;
    for (int g = g_lo; g < g_hi; ++g) {
        const int pi = g % p.np, layer = g / p.np;
        const Phase& ph = p.ph[pi];
        if ((int)blockIdx.x >= ph.parts) continue;
        const char* wsrc = p.W + (size_t)layer * W_LAYER + ph.woff + (size_t)blockIdx.x * ph.wbytes;
        // ---- weights: whatever fits the ring is requested BEFORE the wait for the activations
        int wreq = prefetched_g == g ? prefetched_n : 0;
        if (wreq == 0 && ph.wbytes > 0) { wreq = min(ph.wbytes, LDS_W); stage_weights(wsrc, 0, wreq, ring, wave, lane); }
        // ---- wait for the previous phase
        if (PERSIST && g > 0) {
            if (wave == 0) {
                const unsigned* c = p.cnt + (size_t)(g - 1) * 256;
                const unsigned target = (unsigned)p.ph[(g - 1) % p.np].parts;
                unsigned spins = 0; bool ok = true;
                for (;;) {
                    unsigned v = lane < 8 ? ld_relaxed(c + lane * 32) : 0u;
#pragma unroll
                    for (int m = 4; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
                    if (__shfl(v, 0, 64) >= target) break;
                    if (++spins > SPIN_MAX || ((spins & 255u) == 0u && ld_relaxed(p.err + 1) != 0u)) { ok = false; break; }
                }
                if (!ok && lane == 0) { atomicAdd(p.err + 1, 1u); *s_flag = 1; }
                if (SYNC == 0 && lane == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            }
            __syncthreads();
            if (*s_flag) return;
        }
        // ---- activations in 64-KiB steps through LDS (every word checked), the weight stream continues beside them, MFMA chain per 64-KiB of K
        const char* asrc = p.act[(g + 1) & 1];
        const unsigned epoch_prev = (unsigned)g;
        const int prev_bytes = g > 0 ? p.ph[(g - 1) % p.np].parts * p.ph[(g - 1) % p.np].wr : 0;   // what the previous phase really wrote (the rest of a larger read is traffic only)
        unsigned bad = 0;
        int mf_done = 0;
        const int steps = max(1, (ph.rd + LDS_A - 1) / LDS_A);
        for (int s = 0; s < steps; ++s) {
            const int a0 = s * LDS_A, an = min(LDS_A, ph.rd - a0);
            if (wave >= 1 && an > 0 && g > 0) {
                for (int piece = wave - 1; piece * 1024 < an; piece += 4) {
                    const char* gsrc = asrc + a0 + piece * 1024 + lane * 16;
                    if (SYNC == 1) __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)(abuf + piece * 1024), 16, 0, 16);
                    else __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)(abuf + piece * 1024), 16, 0, 0);
                }
            }
            // weights beyond the ring: issued as the stream advances (a slot is free once its K range has been multiplied)
            if (wreq < ph.wbytes) {
                const int want = min(ph.wbytes, LDS_W + (int)((long long)ph.wbytes * (s + 1) / steps));
                if (want > wreq) { stage_weights(wsrc, wreq, want - wreq, ring, wave, lane); wreq = want; }
            }
            if (wave >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (an > 0 && g > 0)
                for (int i = tid * 16; i < an && a0 + i < prev_bytes; i += THREADS * 16) {
                    const uint4 v = *reinterpret_cast<const uint4*>(abuf + i);
                    bad += (v.x != epoch_prev) + (v.y != epoch_prev) + (v.z != epoch_prev) + (v.w != epoch_prev);
                }
            if (wave == 0) {
                const int mf_to = (int)((long long)ph.nmfma * (s + 1) / steps);
                if (mf_to > mf_done) {
                    f32x16 acc = {};
                    f16x8 a, b;
                    const uint4 wv = *reinterpret_cast<const uint4*>(ring + ((lane * 16 + s * 4096) & (LDS_W - 1)));   // touch the staged weights
                    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f); b[i] = (_Float16)((wv.x >> i) & 1u); }
                    for (int i = mf_done; i < mf_to; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                    if (acc[0] == 12345.678f) atomicAdd(p.err + 2, 1u);
                    mf_done = mf_to;
                }
            }
            __syncthreads();
        }
        if (bad) atomicAdd(p.err, bad);
        // ---- output slice
        const unsigned epoch = (unsigned)(g + 1);
        const u32x4 val = {epoch, epoch, epoch, epoch};
        char* my = p.act[g & 1] + (size_t)blockIdx.x * ph.wr;
        for (int i = tid * 16; i < ph.wr; i += THREADS * 16) {
            if (PERSIST) store_sc1(my, i, val); else *reinterpret_cast<u32x4*>(my + i) = val;
        }
        if (!PERSIST) continue;
        // ---- next participating phase's weights, then publish
        prefetched_g = -1;
        for (int g2 = g + 1; g2 < total && g2 <= g + p.np; ++g2) {
            const Phase& nx = p.ph[g2 % p.np];
            if ((int)blockIdx.x < nx.parts) {
                if (nx.wbytes > 0) {
                    const char* nsrc = p.W + (size_t)(g2 / p.np) * W_LAYER + nx.woff + (size_t)blockIdx.x * nx.wbytes;
                    prefetched_n = min(nx.wbytes, LDS_W); prefetched_g = g2;
                    stage_weights(nsrc, 0, prefetched_n, ring, wave, lane);       // lands while the next phase waits for its activations
                }
                break;
            }
        }
        // the output stores were issued before the prefetch: wait for them only (the prefetch pieces are younger and may stay in flight)
        if (prefetched_g >= 0 && wave >= 1) {
            const int mine = (prefetched_n / 1024 - (wave - 1) + 3) / 4;     // prefetch pieces this wave just issued
            if (mine >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); else if (mine >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.cnt + (size_t)g * 256 + (blockIdx.x & 7) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 7;
    CK(hipSetDevice(0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int G = (prop.multiProcessorCount / 8) * 8;
    Params p{};
    const size_t cnt_bytes = (size_t)LAYERS * MAXP * 256 * 4;
    CK(hipMalloc(&p.cnt, cnt_bytes));
    CK(hipMalloc(&p.act[0], ACT_BYTES)); CK(hipMalloc(&p.act[1], ACT_BYTES));
    char* W; CK(hipMalloc(&W, W_LAYER * LAYERS)); CK(hipMemset(W, 1, W_LAYER * LAYERS)); p.W = W;
    CK(hipMalloc(&p.err, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = LDS_W + LDS_A + 16;
    auto set_lds = [&](auto kern) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); };
    set_lds(&k_model<0, true>); set_lds(&k_model<1, true>); set_lds(&k_model<0, false>);
    printf("# %s, %d CUs: timing model of a 32-token BERT-large forward (24 layers), weights streamed from a %zu-MiB buffer\n", prop.gcnArchName, prop.multiProcessorCount,
           (W_LAYER * LAYERS) >> 20);
    for (int fuse = 0; fuse < 2; ++fuse) {
        const size_t MB = 1 << 20;
        if (!fuse) {
            p.np = 7;
            const Phase ph[7] = {{96, 65536, 65536, 64, 2048, 0}, {16, 0, 12288, 16, 4096, 0}, {32, 65536, 65536, 64, 2048, 6 * MB}, {8, 0, 20480, 0, 12288, 0},
                                 {128, 65536, 65536, 64, 2048, 8 * MB}, {32, 262144, 262144, 256, 2048, 16 * MB}, {8, 0, 20480, 0, 12288, 0}};
            for (int i = 0; i < 7; ++i) p.ph[i] = ph[i];
        } else {
            p.np = 5;     // LayerNorm recomputed by every consumer block of the next GEMM (reads y + residual hi + lo of its 32 rows: 160 KiB instead of 64)
            const Phase ph[5] = {{96, 65536, 163840, 64, 2048, 0}, {16, 0, 12288, 16, 4096, 0}, {32, 65536, 65536, 64, 2048, 6 * MB},
                                 {128, 65536, 163840, 64, 2048, 8 * MB}, {32, 262144, 262144, 256, 2048, 16 * MB}};
            for (int i = 0; i < 5; ++i) p.ph[i] = ph[i];
        }
        printf("## %s: %d phases per layer\n", fuse ? "LayerNorm as the consumer's prologue" : "phases as today", p.np);
        const char* names[3] = {"persistent, sharded counters + acquire", "persistent, sharded counters, sc1 LDS-DMA loads, no acquire", "one launch per phase"};
        for (int mode = 0; mode < 3; ++mode) {
            std::vector<float> ms; unsigned herr[2] = {0, 0};
            for (int r = 0; r < reps; ++r) {
                CK(hipMemsetAsync(p.cnt, 0, cnt_bytes, st)); CK(hipMemsetAsync(p.err, 0, 64, st));
                CK(hipMemsetAsync(p.act[0], 0, ACT_BYTES, st)); CK(hipMemsetAsync(p.act[1], 0, ACT_BYTES, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                if (mode == 0) hipLaunchKernelGGL((k_model<0, true>), dim3(G), dim3(THREADS), lds, st, p);
                else if (mode == 1) hipLaunchKernelGGL((k_model<1, true>), dim3(G), dim3(THREADS), lds, st, p);
                else for (int g = 0; g < LAYERS * p.np; ++g) { p.only = g; hipLaunchKernelGGL((k_model<0, false>), dim3(p.ph[g % p.np].parts), dim3(THREADS), lds, st, p); }
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st)); CK(hipGetLastError());
                float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
                unsigned he[2]; CK(hipMemcpy(he, p.err, 8, hipMemcpyDeviceToHost)); herr[0] += he[0]; herr[1] += he[1];
            }
            std::sort(ms.begin(), ms.end());
            printf("%-64s min %.3f ms  median %.3f ms per forward = %.2f us per layer   stale words %u, timeouts %u\n", names[mode], ms[0], ms[ms.size() / 2],
                   ms[ms.size() / 2] * 1e3 / LAYERS, herr[0], herr[1]);
            fflush(stdout);
        }
    }
    return 0;
}
