// Seam microbenchmark for a single-launch small-batch encoder forward ("if grid barriers cost > 3 us each on this stack, measure
// that first").  It emulates the DEPENDENCY CHAIN of one BERT layer at 32 tokens without its arithmetic: a sequence of phases
//     QKV (96 tiles) -> attention (16 heads) -> out-proj (32) -> LN (8) -> FF1 (128) -> FF2 (32) -> LN (8)
// where phase p's participants (workgroups 0 .. parts[p]-1 of a resident 256-workgroup grid) wait for ALL participants of phase p-1, read rd[p] bytes
// of what they wrote (LDS-DMA into LDS, every byte checked against the expected epoch: a stale line is an error, not a slow-down), run a dependent
// MFMA chain of nmfma[p] instructions (the K = 1024 / 4096 accumulator chain of one 32x32 tile), write wr[p] bytes and signal.
// Hand-off forms (cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "visibility"):
//   mode 0  R1: payload stored write-through (sc1), every storing wave drains, one lane adds to the phase counter; consumer: one lane polls (relaxed, sc1),
//           ONE agent acquire (buffer_inv sc1), vmcnt(0), workgroup barrier, plain LDS-DMA loads
//   mode 1  plain stores + agent release fence + counter add; consumer as in mode 0
//   mode 2  the same bodies as ONE LAUNCH PER PHASE (the kernel boundary is the seam): what the present forward does
//   mode 3  R1 producer, consumer WITHOUT the acquire, LDS-DMA loads carry sc1 (aux 16) instead: unmeasured by the guide; the byte check says whether it holds here
//   mode 4  full grid barrier between phases (every workgroup arrives, XCD-hierarchical counters), R1 payload
// Build: hipcc -O3 --offload-arch=gfx950 tools/seam_bench.hip -o tools/bin/seam_bench ; run on the GPU box: tools/bin/seam_bench [layers] [reps]
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

constexpr int NP = 7;
constexpr int THREADS = 320;          // the skinny loop's geometry: 1 multiplying + 4 staging waves
constexpr int BUF_BYTES = 256 * 1024;
constexpr unsigned SPIN_MAX = 4000000u;

struct Params {
    unsigned* cnt;       // [layers * NP] arrival counters (mode 4: [layers * NP][16]: 8 per-XCD + 1 top + generation words)
    char* buf[2];        // 2 x 256 KiB, phase parity
    unsigned* err;       // [0] stale words seen, [1] timeouts
    int layers;
    int parts[NP], rd[NP], wr[NP], nmfma[NP];
    int mode;
    int only_phase;      // mode 2: global phase index this launch runs
    int G;
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// returns false on timeout
__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned target, unsigned* err) {
    unsigned spins = 0;
    while (ld_relaxed(p) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > SPIN_MAX || ld_relaxed(err + 1) != 0u) { atomicAdd(err + 1, 1u); return false; }
    }
    return true;
}

// sc1 16-byte store (write-through): raw buffer store with aux = 16
__device__ __forceinline__ void store_sc1(char* base, int off, u32x4 v) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);
}

template <int MODE>
__device__ __forceinline__ void phase_body(const Params& p, int g, int ph, char* smem, bool& dead) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned epoch_prev = (unsigned)g;          // epoch written by global phase g-1 is g (phase index + 1)
    const char* src = p.buf[(g + 1) & 1];
    char* dst = p.buf[g & 1];
    // ---- read rd bytes of the previous phase's output through LDS (64-KiB chunks), check every word
    if (g > 0) {
        unsigned bad = 0;
        for (int base = 0; base < p.rd[ph]; base += 65536) {
            const int n = min(65536, p.rd[ph] - base);
            if (base > 0) __syncthreads();
            if (wave >= 1) {
                for (int piece = wave - 1; piece * 1024 < n; piece += 4) {
                    const char* gsrc = src + base + piece * 1024 + lane * 16;
                    if constexpr (MODE == 3) __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)(smem + piece * 1024), 16, 0, 16);
                    else __builtin_amdgcn_global_load_lds((gbl_void*)gsrc, (lds_void*)(smem + piece * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            for (int i = tid * 16; i < n; i += THREADS * 16) {
                const uint4 v = *reinterpret_cast<const uint4*>(smem + i);
                bad += (v.x != epoch_prev) + (v.y != epoch_prev) + (v.z != epoch_prev) + (v.w != epoch_prev);
            }
        }
        if (bad) atomicAdd(p.err, bad);
    }
    // ---- the accumulator chain of one 32x32 tile
    if (wave == 0 && p.nmfma[ph] > 0) {
        f32x16 acc = {};
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f); b[i] = (_Float16)(i * 0.002f); }
        for (int i = 0; i < p.nmfma[ph]; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (acc[0] == 12345.678f) atomicAdd(p.err, 1u);     // keep the chain
    }
    // ---- write wr bytes (this workgroup's slice) with the new epoch
    const unsigned epoch = (unsigned)(g + 1);
    const u32x4 val = {epoch, epoch, epoch, epoch};
    char* my = dst + (size_t)blockIdx.x * p.wr[ph];
    for (int i = tid * 16; i < p.wr[ph]; i += THREADS * 16) {
        if constexpr (MODE == 1 || MODE == 2) *reinterpret_cast<u32x4*>(my + i) = val;
        else store_sc1(my, i, val);
    }
}

template <int MODE>
__global__ __launch_bounds__(THREADS) void k_chain(Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    __shared__ int s_dead;
    if (tid == 0) s_dead = 0;
    __syncthreads();
    const int total = p.layers * NP;
    const int g_lo = MODE == 2 ? p.only_phase : 0, g_hi = MODE == 2 ? p.only_phase + 1 : total;
    for (int g = g_lo; g < g_hi; ++g) {
        const int ph = g % NP;
        const bool part = (int)blockIdx.x < p.parts[ph];
        if constexpr (MODE == 4) {
            // full grid barrier before every phase but the first: per-XCD counter -> top counter -> per-XCD generation word
            if (g > 0) {
                unsigned* c = p.cnt + (size_t)(g - 1) * 32;
                const int x = blockIdx.x & 7;
                const unsigned per_x = (unsigned)(p.G / 8);
                if (tid == 0) {
                    const unsigned old = __hip_atomic_fetch_add(c + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old == per_x - 1) {
                        const unsigned t = __hip_atomic_fetch_add(c + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (t == 7u) for (int i = 0; i < 8; ++i) __hip_atomic_store(c + 16 + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (!wait_ge(c + 16 + x, 1u, p.err)) s_dead = 1;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                if (s_dead) return;
            }
        } else if constexpr (MODE != 2) {
            if (part && g > 0) {
                if (tid == 0) {
                    if (!wait_ge(p.cnt + (g - 1), (unsigned)p.parts[(g - 1) % NP], p.err)) s_dead = 1;
                    if constexpr (MODE != 3) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                }
                __syncthreads();
                if (s_dead) return;
            }
        }
        if (!part) continue;
        bool dead = false;
        phase_body<MODE>(p, g, ph, smem, dead);
        // ---- publish
        if constexpr (MODE == 2) continue;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains
        __syncthreads();
        if (tid == 0) {
            if constexpr (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if constexpr (MODE != 4) __hip_atomic_fetch_add(p.cnt + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    (void)lane; (void)wave;
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 24;
    const int reps = argc > 2 ? atoi(argv[2]) : 7;
    CK(hipSetDevice(0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int G = (prop.multiProcessorCount / 8) * 8;
    printf("# %s, %d CUs, grid %d x %d threads, %d layers x %d phases\n", prop.gcnArchName, prop.multiProcessorCount, G, THREADS, layers, NP);
    Params p{};
    p.layers = layers; p.G = G;
    const int parts[NP] = {96, 16, 32, 8, 128, 32, 8};
    const int rd[NP] = {65536, 12288, 65536, 8192, 65536, 262144, 8192};
    const int wr[NP] = {2048, 4096, 2048, 8192, 2048, 2048, 8192};
    const int nm[NP] = {64, 16, 64, 0, 64, 256, 0};
    for (int i = 0; i < NP; ++i) { p.parts[i] = parts[i]; p.rd[i] = rd[i]; p.wr[i] = wr[i]; p.nmfma[i] = nm[i]; }
    const size_t cnt_bytes = (size_t)layers * NP * 32 * 4;
    CK(hipMalloc(&p.cnt, cnt_bytes));
    CK(hipMalloc(&p.buf[0], BUF_BYTES)); CK(hipMalloc(&p.buf[1], BUF_BYTES));
    CK(hipMalloc(&p.err, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = 65536;
    auto set_lds = [&](auto kern) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds)); };
    set_lds(&k_chain<0>); set_lds(&k_chain<1>); set_lds(&k_chain<2>); set_lds(&k_chain<3>); set_lds(&k_chain<4>);
    const char* names[5] = {"R1 sc1 payload + acquire (per-phase counters)", "plain stores + release fence + acquire", "one launch per phase (kernel boundary)",
                            "R1 + sc1 LDS-DMA loads, NO acquire", "full grid barrier per phase (XCD-hierarchical), R1 payload"};
    for (int variant = 0; variant < 2; ++variant) {
        // variant 1: the bare chain — no payload reads, no MFMA chain, 16-byte writes: the cost of the synchronisation alone
        if (variant == 1) for (int i = 0; i < NP; ++i) { p.rd[i] = 0; p.nmfma[i] = 0; p.wr[i] = 16; }
        printf("## %s\n", variant == 0 ? "layer emulation at 32 tokens (payload reads + MFMA chains)" : "synchronisation alone (no payload, no arithmetic)");
        for (int mode = 0; mode < 5; ++mode) {
            std::vector<float> ms;
            unsigned herr[2] = {0, 0};
            for (int r = 0; r < reps; ++r) {
                CK(hipMemsetAsync(p.cnt, 0, cnt_bytes, st));
                CK(hipMemsetAsync(p.err, 0, 64, st));
                CK(hipMemsetAsync(p.buf[0], 0, BUF_BYTES, st)); CK(hipMemsetAsync(p.buf[1], 0, BUF_BYTES, st));
                CK(hipStreamSynchronize(st));
                p.mode = mode;
                CK(hipEventRecord(e0, st));
                if (mode == 2) {
                    for (int g = 0; g < layers * NP; ++g) { p.only_phase = g; hipLaunchKernelGGL(k_chain<2>, dim3(p.parts[g % NP]), dim3(THREADS), lds, st, p); }
                } else if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(G), dim3(THREADS), lds, st, p);
                else if (mode == 1) hipLaunchKernelGGL(k_chain<1>, dim3(G), dim3(THREADS), lds, st, p);
                else if (mode == 3) hipLaunchKernelGGL(k_chain<3>, dim3(G), dim3(THREADS), lds, st, p);
                else hipLaunchKernelGGL(k_chain<4>, dim3(G), dim3(THREADS), lds, st, p);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipGetLastError());
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t);
                unsigned he[2]; CK(hipMemcpy(he, p.err, 8, hipMemcpyDeviceToHost));
                herr[0] += he[0]; herr[1] += he[1];
            }
            std::sort(ms.begin(), ms.end());
            printf("mode %d  %-62s  min %.3f ms  median %.3f ms  = %.2f us per layer, %.2f us per seam   stale words %u, timeouts %u\n", mode, names[mode], ms[0],
                   ms[ms.size() / 2], ms[ms.size() / 2] * 1e3 / layers, ms[ms.size() / 2] * 1e3 / (layers * NP), herr[0], herr[1]);
            fflush(stdout);
        }
    }
    return 0;
}
