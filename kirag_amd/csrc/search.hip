// Flat inner-product index on MI355X — replaces faiss.IndexFlatIP behind retriever/index.py (reference
// Indexer.index_data :26-34, Indexer.search_knn :36-53).
//
// Data in HBM, per row: fp32 master [n, d] (exact re-rank) + 16-bit copy [n, dpad] (coarse MFMA scan).
//
// search(q[nq,d], k), per block of <= 1024 queries (the reference's index_batch_size, index.py:39-46):
//   1. k_prep_queries   q -> 16-bit Qc (zero-padded to 128 queries), eps_q = rigorous bound on
//                       |canonical(q,x) - coarse(q,x)| over all stored rows x
//   2. rounds r = 0..R-1 over geometrically growing, interleaved subsets of the 128-row corpus tiles:
//        k_coarse   Qc x Xc^T by MFMA; every score >= thr[q] is appended to the query's candidate buffer
//                   (the [nq, N] score matrix never exists)
//        k_select   sort the buffer; keep the K1 best; thr[q] = K1-th best (a valid lower bound of the
//                   final K1-th best because it is the K1-th best of a subset)
//   3. k_rerank   b_k = k-th best coarse score; every row of the exact top-k has coarse score
//                 >= theta = b_k - 2*eps_q; the buffer holds ALL rows with coarse score >= thr, so if
//                 theta > thr the exact answer is inside the buffer prefix {coarse >= theta}: gather those
//                 fp32 rows, compute the canonical fp64-ordered score (oracle/search_c.c), sort by
//                 (score desc, row asc), emit k.  Otherwise (or on buffer overflow) flag the query.
//   4. flagged queries are re-answered by the fp64-MFMA scan of the fp32 rows (k_fine, pass 2) and, failing that, by the exact full
//      scan (k_exact_scan + k_sort_chunks tree, pass 3).
// Blocks of <= 32 queries on a large index stream a derived int8 copy (1 byte per element + one scale per row) in the LAST round of
// step 2 and 16-bit-score only the rows it marks (byte_final_round).
// The result is therefore ALWAYS the exact top-k under the canonical score; the low-precision scans only decide
// which rows get the fp64 treatment.
#include "gemm_nt.hpp"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <type_traits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <functional>
#include <vector>

namespace kr {

using ShapeC = ShapePP;                     // coarse scan tile: 256 corpus rows x 256 queries, 8 waves of 128x64, ping-pong main loop (128 KiB LDS ring)
constexpr int COARSE_STAGES = 2;
constexpr int QBLK = 1024;                   // queries per search block (reference index_batch_size)
constexpr int EXACT_RC = 1024;               // rows per block of the exact scan
constexpr int SORT_CHUNK = 4096;             // keys per block of the merge tree
constexpr int THETA_BLOCKS = 16;             // pass 1's theta is kept for the first 16 query blocks of a call (later blocks: pass 2 without the pre-scan)
constexpr int TIMED_BLOCKS = 4;              // blocks of a call whose coarse rounds are bracketed by events
constexpr int STATUS_STRIDE = 2 * QBLK + 4;  // per-block status record in pinned memory: [QBLK] flags | [QBLK] re-ranked rows | [4] list overflow

// Growable device array without copies: a virtual address range reserved once (sized for the whole HBM: an index can never need more) into which
// physical chunks are mapped as the row count grows (hipMemAddressReserve / hipMemCreate / hipMemMap).  The rows never move, so appending to a
// 100-GB index needs neither a second allocation nor a device-to-device copy (round 1: grow() = hipMalloc of 1.5x + full copy, a transient 2.5x
// footprint whenever the caller had not called reserve()).
static std::atomic<unsigned long long> g_va_retired{0};   // bytes of virtual address space retired by released VBufs (see VBuf::release)
struct VBuf {
    char* base = nullptr;
    size_t reserved = 0, mapped = 0, chunk = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    int device = 0;

    int init(int dev, size_t reserve_bytes) {
        device = dev;
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) { (void)hipGetLastError(); return -1; }
        chunk = ((size_t)(64u << 20) + gran - 1) / gran * gran;
        reserved = (reserve_bytes + chunk - 1) / chunk * chunk;
        void* p = nullptr;
        if (hipMemAddressReserve(&p, reserved, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); reserved = 0; return -1; }
        base = static_cast<char*>(p);
        return 0;
    }
    // make [0, bytes) backed by memory; 0 on success, -1 = this mechanism is unavailable / out of range (nothing changed), KR_ENOMEM = out of memory
    int ensure(size_t bytes) {
        if (bytes <= mapped) return 0;
        if (!base || bytes > reserved) return -1;
        hipMemAllocationProp prop{};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        while (mapped < bytes) {
            hipMemGenericAllocationHandle_t h;
            hipError_t e = hipMemCreate(&h, chunk, &prop, 0);
            if (e != hipSuccess) { (void)hipGetLastError(); return e == hipErrorOutOfMemory ? KR_ENOMEM : -1; }
            if (hipMemMap(base + mapped, chunk, 0, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipMemRelease(h); return -1; }
            if (hipMemSetAccess(base + mapped, chunk, &acc, 1) != hipSuccess) { (void)hipGetLastError(); (void)hipMemUnmap(base + mapped, chunk); (void)hipMemRelease(h); return -1; }
            handles.push_back(h);
            mapped += chunk;
        }
        return 0;
    }
    void release() {
        for (size_t i = 0; i < handles.size(); ++i) { (void)hipMemUnmap(base + i * chunk, chunk); (void)hipMemRelease(handles[i]); }
        handles.clear(); mapped = 0;
        // The address range is deliberately NOT handed back (hipMemAddressFree): on ROCm 7.2 / gfx950 a range that is freed, reserved again (the runtime
        // returns the same addresses) and mapped to new physical chunks is read by kernels through stale translations — the fp32 rows a copy engine
        // reads back are correct while the scan kernels see other memory (found by tests/soak_gpu.py; tools/vmm_va_reuse_repro.py, fixed when the range
        // is never reused).  Physical memory IS returned (unmap + release above); only virtual addresses are retired, counted in g_va_retired, and
        // grow() stops using this path before the 47-bit address space could run short.
        if (base) {
            if (getenv("KIRAG_AMD_DEBUG_FREE_VA")) (void)hipMemAddressFree(base, reserved);   // the broken behaviour, for the repro tool only
            else g_va_retired.fetch_add(reserved);
        }
        base = nullptr; reserved = 0;
    }
};

struct Index {
    int d = 0, dpad = 0, coarse = KR_COARSE_BF16, device = 0;
    VBuf vf, vc;               // backing of xf / xc when the virtual-memory path is in use (vmm == 1)
    int vmm = -1;              // -1 not decided yet, 0 hipMalloc + copy-on-grow, 1 mapped chunks
    int64_t n = 0, cap_rows = 0;
    float* xf = nullptr;       // [cap_rows, d] fp32 master
    uint16_t* xc = nullptr;    // [cap_rows, dpad] 16-bit coarse copy
    float* bounds = nullptr;   // [2] device: max_x ||x - c(x)||_2 , max_x ||c(x)||_2   (non-negative -> uint order)
    // search workspace (sized for QBLK queries)
    float* q_f = nullptr;      // [QBLK, d]
    float* q_f2 = nullptr;     // [FINE_QMAX, d] compacted flagged queries of the high-precision pass
    float* theta1 = nullptr;   // [THETA_BLOCKS, QBLK] pass 1's b_k - 2 eps per query (k_rerank)
    float* thr_mark = nullptr; // [32] the same, compacted for the group being pre-scanned
    uint32_t* bitmap = nullptr; size_t bitmap_words = 0;   // one bit per row (+ one word: the list length) — pass 2 pre-scan
    size_t cnt_word_at = (size_t)-1;   // where the list-length word behind the bits was last put (index (n + 31) / 32 at that time): it holds a count afterwards, and the
                                       // bitmap is allocated for more rows than the index has, so it is zeroed when the row count moves it (move_count_word)
    uint32_t* rowlist = nullptr;                           // the marked rows, compacted
    uint32_t* qmask = nullptr;                             // byte pre-scan: per row, the queries of the block that marked it (all-zero between searches)
    // byte pre-scan of small query blocks (see byte_final_round): an int8 copy of the rows with one scale per row.  A DERIVED structure: built lazily from
    // xf by the first small-block search (ensure_byte_copy), extended behind later adds, never part of the growth machinery above
    int8_t* x8 = nullptr;      // [cap8, dpad8]
    float* sx8 = nullptr;      // [cap8] row scale: x ~ sx8[row] * x8[row]   (NaN for the padding rows behind n8)
    float* bounds8 = nullptr;  // [4] device: max ||r - sx8 x8||_2 over the rows (r = (x - mu) w) ; word 1: != 0 once a row with a non-finite element has been seen ; max ||r||_2
    int8_t* q8 = nullptr;      // [2][32][dpad8] the block's queries as two byte planes: q ~ sq (q8[0] + q8[1] / 254)
    float* thr8 = nullptr;     // [32] per-query mark threshold in units of the integer score (k_scan8_prep)
    uint32_t* cnt_spread = nullptr;   // [32 * CNT_STRIDE] the block's candidate counters, one per 4 KiB, while k_score_list appends (see there)
    float* mu8 = nullptr;      // [3][dpad8] centre mu, axis weights w, 1 / w (k_mu_final; fixed for the life of the copy), then the partial sums
    int64_t n8 = 0, cap8 = 0; int dpad8 = 0;
    bool warmed = false;       // kr_index_prepare has run its one warm-up search
    bool byte_off = false;     // environment switch / allocation failure / non-finite rows: the 16-bit scan serves every block
    int byte_bad = 0, byte_pause = 0;   // feedback from finished calls: pre-scans that marked too many rows in a row; calls left without a pre-scan
    int force_exact = 0;       // test hook: every canonical score through the integer super-accumulator
    uint16_t* q_c = nullptr;   // [QBLK, dpad]
    float* thr = nullptr;      // [QBLK]
    float* eps = nullptr;      // [QBLK]
    uint32_t* cnt = nullptr;   // [QBLK]
    uint32_t* flags = nullptr; // [QBLK] bit0 overflow, bit1 uncertified
    uint64_t* cand = nullptr;  // [QBLK, cand_cap]
    int cand_cap = 0;
    float* out_s = nullptr;    // [QBLK, kmax]
    int64_t* out_r = nullptr;  // [QBLK, kmax]
    uint32_t* nrer = nullptr;  // [QBLK] re-ranked rows (stats)
    uint32_t* h_status = nullptr;   // pinned host: pass 2: [QBLK] flags | [QBLK] nrer | [QBLK] group list overflow | [16] marked rows; then one STATUS_STRIDE record per query block
    int out_k = 0;
    uint64_t* ex_a = nullptr; uint64_t* ex_b = nullptr; size_t ex_bytes = 0;  // exact-scan ping/pong
    int* ex_qidx = nullptr;    // [QBLK] flagged query list
    uint4* blk_list = nullptr; unsigned int* blk_cnt = nullptr;   // [num_cu*8, WLISTCAP] per-wave survivor lists, [num_cu*8 + 1] counts (+ overflow word)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t evc[TIMED_BLOCKS * 2 * 16] = {};   // begin/end pairs around each coarse round of the first blocks of a call (roofline timing)
    hipEvent_t ev_add = nullptr;   // recorded behind the last (asynchronous) add: searches on another stream wait for it
    kr_search_stats st{};
    int num_cu = 256;
    // environment switches, read ONCE at kr_index_create (round 2 called getenv() on every search)
    bool no_q32 = false, no_fine = false, no_mark = false, no_vmm = false;
    // asynchronous search (kr_index_search_async ... kr_index_search_finish): pass 1 of every block is enqueued, the per-query certificate flags
    // land in pinned memory behind it; finish() reads them and runs the rare passes 2 / 3
    // Up to PEND_MAX calls may be outstanding on ONE stream (the row-sharded search enqueues the W batches of a block back to back and looks at their
    // certificates once): every call owns a slot with its own pinned status records and completion event; the device workspace is shared in stream order.
    struct Pending {
        bool active = false;
        const float* q = nullptr; int nq = 0, k = 0; float* scores = nullptr; int64_t* rows = nullptr; hipStream_t st = nullptr;
        std::vector<int> rounds;        // coarse rounds per block (event pairs to read)
        uint32_t* status = nullptr; int status_blocks = 0;   // pinned host: one STATUS_STRIDE record per query block of THIS call
        hipEvent_t ev_done = nullptr;   // behind the last enqueued block of the call
        uint64_t seq = 0;               // the call's number (the workspace-resident theta1 / timing events belong to the newest call only)
        bool half = false;              // kr_index_search_coarse_async was enqueued, kr_index_search_rerank_async not yet
        int final_preset = 0, rmax = 0; // ... what the second half needs from the first
        bool byte_used = false;
        bool direct_io = false;         // pass 1 read the queries / wrote the results in the caller's own buffers: the workspace holds neither
    };
    static constexpr int PEND_MAX = 16;
    Pending pend[PEND_MAX];
    int pend_head = 0, pend_n = 0;  // ring of outstanding calls: slots pend_head .. pend_head + pend_n - 1 (mod PEND_MAX), oldest first
    uint64_t call_seq = 0;
    uint64_t ws_owner = 0;          // the call whose last query block (queries, pass-1 results) the search workspace currently holds
    int status_blocks = 0;          // h_status: the pass-2 status region exists
};

// ---------------------------------------------------------------------------------------------------------
// add: fp32 rows -> 16-bit copy + quantisation-error bounds
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_add_rows(const float* __restrict__ xf, uint16_t* __restrict__ xc, int64_t n, int d, int dpad,
                                                  float* __restrict__ bounds) {
    // grid-stride over rows (one wave per row) with the two maxima kept in registers: ONE pair of atomics per wave at the end.  (With a pair per
    // row the 2 x 250k atomics on one cache line serialised in L2: 5.7 ms per 250k-row add instead of the 0.4 ms the 1.5 GB of traffic need.)
    const int lane = threadIdx.x & 63;
    float emax = 0.f, cmax = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * 4) {
        const float* src = xf + row * d;
        uint16_t* dst = xc + row * dpad;
        float e2 = 0.f, c2 = 0.f;
        for (int i = lane * 4; i < dpad; i += 256) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < d) v = *reinterpret_cast<const float4*>(src + i);
            ushort4 o;
            o.x = T::from_f32(v.x); o.y = T::from_f32(v.y); o.z = T::from_f32(v.z); o.w = T::from_f32(v.w);
            const float cx = T::to_f32(o.x), cy = T::to_f32(o.y), cz = T::to_f32(o.z), cw = T::to_f32(o.w);
            e2 += (v.x - cx) * (v.x - cx) + (v.y - cy) * (v.y - cy) + (v.z - cz) * (v.z - cz) + (v.w - cw) * (v.w - cw);
            c2 += cx * cx + cy * cy + cz * cz + cw * cw;
            *reinterpret_cast<ushort4*>(dst + i) = o;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { e2 += __shfl_xor(e2, m, 64); c2 += __shfl_xor(c2, m, 64); }
        // slack factor covers the fp32 rounding of these sums themselves; NaN rows (never retrievable) are skipped
        const float e = sqrtf(e2) * 1.0001f, c = sqrtf(c2) * 1.0001f;
        if (e == e) emax = fmaxf(emax, e);
        if (c == c) cmax = fmaxf(cmax, c);
    }
    if (lane == 0) {   // non-negative floats order like their bit patterns
        atomicMax(reinterpret_cast<unsigned int*>(bounds), __float_as_uint(emax));
        atomicMax(reinterpret_cast<unsigned int*>(bounds) + 1, __float_as_uint(cmax));
    }
}

// queries: fp32 -> 16-bit (rows >= nq zero), eps_q, and state reset
template <class T>
__global__ __launch_bounds__(64) void k_prep_queries(const float* __restrict__ qf, uint16_t* __restrict__ qc, int nq, int d, int dpad,
                                                     const float* __restrict__ bounds, float* __restrict__ eps, float* __restrict__ thr,
                                                     uint32_t* __restrict__ cnt, uint32_t* __restrict__ flags, unsigned int* __restrict__ zero = nullptr, int nzero = 0) {
    const int q = blockIdx.x, lane = threadIdx.x;
    for (int i = q * 64 + lane; i < nzero; i += (int)gridDim.x * 64) zero[i] = 0u;     // the wave-list counters of the scans behind this kernel
    uint16_t* dst = qc + (int64_t)q * dpad;
    float q2 = 0.f, e2 = 0.f, c2 = 0.f;
    for (int i = lane * 4; i < dpad; i += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq && i < d) v = *reinterpret_cast<const float4*>(qf + (int64_t)q * d + i);
        ushort4 o;
        o.x = T::from_f32(v.x); o.y = T::from_f32(v.y); o.z = T::from_f32(v.z); o.w = T::from_f32(v.w);
        const float cx = T::to_f32(o.x), cy = T::to_f32(o.y), cz = T::to_f32(o.z), cw = T::to_f32(o.w);
        q2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        e2 += (v.x - cx) * (v.x - cx) + (v.y - cy) * (v.y - cy) + (v.z - cz) * (v.z - cz) + (v.w - cw) * (v.w - cw);
        c2 += cx * cx + cy * cy + cz * cz + cw * cw;
        *reinterpret_cast<ushort4*>(dst + i) = o;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { q2 += __shfl_xor(q2, m, 64); e2 += __shfl_xor(e2, m, 64); c2 += __shfl_xor(c2, m, 64); }
    if (lane == 0) {
        const float max_ex = bounds[0], max_cx = bounds[1];
        const float qn = sqrtf(q2) * 1.0001f, qe = sqrtf(e2) * 1.0001f, qcn = sqrtf(c2) * 1.0001f;
        // |q.x - qc.xc| <= |q|.|x - xc| + |q - qc|.|xc|          (Cauchy-Schwarz, exact arithmetic)
        //  + fp32 accumulation of the MFMA chain <= 1.1 * dpad * 2^-24 * |qc|.|xc|
        //  + the final fp32 rounding of the canonical score and of the coarse score (2^-22 * |q|.|x|, |x| <= |xc| + |x-xc|)
        const float acc = 1.1f * (float)dpad * 5.9604645e-8f;
        float e = qn * max_ex + qe * max_cx + acc * qcn * max_cx + 2.4e-7f * qn * (max_cx + max_ex);
        e = e * 1.001f + 1e-30f;
        eps[q] = (q < nq) ? e : 0.f;
        thr[q] = (q < nq) ? -INFINITY : INFINITY;   // padded queries never emit
        cnt[q] = 0u;
        flags[q] = 0u;
    }
}

// ---------------------------------------------------------------------------------------------------------
// coarse scan: Qc x Xc^T, threshold filter in the epilogue
// ---------------------------------------------------------------------------------------------------------
struct CoarseArgs {
    const uint16_t* xc; int64_t n; int dpad;
    const uint16_t* qc; int nq_pad;
    const float* thr; uint32_t* cnt; uint32_t* flags; uint64_t* cand; int cand_cap;
    int nq; int direct;
    uint4* blk_list; unsigned int* blk_cnt; unsigned int* list_overflow;
    int64_t tile_begin, tile_count;   // this round covers permuted tile slots [tile_begin, tile_begin + tile_count)
    int64_t ntiles, perm_mul;         // slot -> tile = (slot * perm_mul) % ntiles   (perm_mul coprime to ntiles)
    const float* xf; const float* qf; int d;   // high-precision pass (k_fine): fp32 master rows, compacted fp32 queries
    uint32_t* bitmap;                          // one bit per ROW, set by the marking scan (k_coarse_q32 MODE 2)
    const uint32_t* rowlist; int64_t nlist;    // k_fine: scan rows rowlist[0 .. nlist) instead of 0 .. n (nullptr: every row)
    const uint16_t* qc2; const float* sx8;     // byte pre-scan (k_coarse_q32 MODE 3): second byte plane of the queries, row scales
    uint32_t* qmask;                           // ... and per row the set of queries that marked it
};

// persistent streaming coarse scan (gemm_nt_pingpong): grid = one block per CU, so nothing else on the CU hides an epilogue
// stall, and the whole kernel must stay inside the instruction cache: the epilogue is branch-light and touches no global
// memory that it has to wait for.
//   direct != 0 (round 0, thr = -inf): every score is stored at slot = (tile slot in round)*BM + row in tile, no atomics.
//   otherwise: thresholds sit in LDS (loaded once per launch); survivors of one accumulator register are compacted with
//   ballot/mbcnt and stored (fire-and-forget) to THIS WAVE's private list, whose cursor lives in an SGPR; scatter_wave_lists (end of the kernel)
//   moves the lists into the per-query buffers.
constexpr int WLISTCAP = 8192;                      // entries per wave list (expected nq*cap/2/(8*num_blocks) ~ 1k-2k)
constexpr int COARSE_LDS = COARSE_STAGES * ShapeC::STAGE_BYTES + QBLK * 4;
constexpr int COARSE_LDS_SMALLQ = SPLIT_RING * ShapeSplit::STAGE_BYTES + QBLK * 4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

// fused scatter: this block's 8 wave lists -> the per-query candidate buffers.  One global atomic per (block, query) reserves a range (instead of one
// per survivor), the position inside the range comes from an LDS counter.  Called by every thread of the block after the main loop (the LDS ring is free).
__device__ __forceinline__ void scatter_wave_lists(const CoarseArgs& a, char* smem, int wave_id, unsigned int wcnt, int nthreads) {
    constexpr int NW = ShapeC::NWAVE;                                    // list slots per block (waves without accumulators have empty lists)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's list stores have reached L2
    unsigned int* hist = reinterpret_cast<unsigned int*>(smem);          // [QBLK]
    unsigned int* base = hist + QBLK;                                    // [QBLK]
    unsigned int* wc = base + QBLK;                                      // [NW]
    __syncthreads();
    if ((int)threadIdx.x < NW) wc[threadIdx.x] = 0u;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        wc[wave_id] = wcnt < (unsigned)WLISTCAP ? wcnt : (unsigned)WLISTCAP;
        if (wcnt > (unsigned)WLISTCAP) *a.list_overflow = 1u;
    }
    for (int i = threadIdx.x; i < a.nq_pad; i += nthreads) hist[i] = 0u;
    __syncthreads();
    const uint4* lists = a.blk_list + (int64_t)blockIdx.x * NW * WLISTCAP;
    for (int w = 0; w < NW; ++w)
        for (unsigned i = threadIdx.x; i < wc[w]; i += nthreads) atomicAdd(&hist[lists[(int64_t)w * WLISTCAP + i].z], 1u);
    __syncthreads();
    for (int q = threadIdx.x; q < a.nq_pad; q += nthreads) {
        const unsigned c = hist[q];
        base[q] = c ? atomicAdd(&a.cnt[q], c) : 0u;
        hist[q] = 0u;
    }
    __syncthreads();
    for (int w = 0; w < NW; ++w)
        for (unsigned i = threadIdx.x; i < wc[w]; i += nthreads) {
            const uint4 e = lists[(int64_t)w * WLISTCAP + i];
            const unsigned pos = base[e.z] + atomicAdd(&hist[e.z], 1u);
            if (pos < (unsigned)a.cand_cap) a.cand[(int64_t)e.z * a.cand_cap + pos] = make_key(__uint_as_float(e.x), e.y);
        }
}

// The 16-bit copy is padded with NaN rows up to a multiple of 256 (k_pad_nan): a partial last tile then needs no row test,
// because NaN scores fail every `>=` and sort below every real key.
// SMALLQ (at most 128 queries in the block: the KiRAG loop's 1-2 queries per hop, single-question retrieval): 128-row x 128-query tiles on the
// producer / consumer loop (gemm_nt_split) instead of 256 x 256 on the ping-pong loop.  With a 256-query tile a small batch pays the MFMA time of 256
// queries (2.0 ms per 5M rows, above the 1.3-1.6 ms the corpus needs to cross HBM); with 128 the scan is HBM-bound.
// (Round 4's diagnostic epilogue variants — survivors as the likely branch, counted but not stored, no filter at all = the 4.8 % ceiling of any epilogue
// re-scheduling — are recorded in profiles/r04/tried_coarse_epilogue_ceiling.txt; their code lived behind -DKR_EXPERIMENT up to commit 81642ee.)
template <class T, bool DIRECT, bool SMALLQ = false>
__global__ __launch_bounds__(512, SMALLQ ? 1 : 2) void k_coarse(CoarseArgs a) {
    using S = std::conditional_t<SMALLQ, ShapeSplit, ShapeC>;      // S::NWAVE = waves that own accumulators (4 of the 8 with SMALLQ)
    constexpr int RING_BYTES = SMALLQ ? SPLIT_RING * ShapeSplit::STAGE_BYTES : COARSE_STAGES * ShapeC::STAGE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* thr_s = reinterpret_cast<float*>(smem + RING_BYTES);
    if (!DIRECT) {
        for (int i = threadIdx.x; i < a.nq_pad; i += ShapeC::NTHREADS) thr_s[i] = a.thr[i];
        __syncthreads();
    }
    const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this wave's survivor list as a buffer resource: 32-bit offsets, and stores past the capacity are dropped by the hardware
    const __amdgpu_buffer_rsrc_t wlist = __builtin_amdgcn_make_buffer_rsrc(
        a.blk_list + ((int64_t)blockIdx.x * ShapeC::NWAVE + wave_id) * WLISTCAP, 0, WLISTCAP * 16, 0x00020000);
    unsigned int wcnt = 0;   // wave-uniform cursor into the list
    const int64_t tn_count = a.nq_pad / S::BN;
    const int64_t total = a.tile_count * tn_count;
    const int64_t n_pad = (a.n + S::BM - 1) / S::BM * S::BM;
    // corpus loads are non-temporal: the corpus is a once-through stream (shared by the 4 CUs of an XCD that hold the other query tiles of the
    // same rows at about the same time), while the 2-MiB query block is re-read by every job; with default-policy corpus loads the stream
    // pushed the queries out of the 4-MiB L2 once per job (FETCH_SIZE 1.54x the corpus bytes; 1.18x with nt, profiles/)
    auto coord =         [&](int64_t nat, int64_t& m0, int64_t& n0) {
            uint32_t tn, tile; uint64_t tslot, qq;                        // the query blocks of one corpus tile are adjacent
            // no 64-bit integer divisions (~150 VALU each, per tile, per wave); the fp64 form is exact below 2^53, which covers a 2^32-row shard
            fast_divmod64((uint64_t)nat, (uint32_t)tn_count, tslot, tn);
            fast_divmod64((uint64_t)(a.tile_begin + (int64_t)tslot) * (uint64_t)a.perm_mul, (uint32_t)a.ntiles, qq, tile);
            m0 = (int64_t)tile * S::BM; n0 = (int64_t)tn * S::BN;
        };
    auto epi = [&](AccTile<S>& acc, int64_t m0, int64_t n0, int64_t nat) {
            const int lane_row0 = acc.m_wave + 4 * (acc.lane >> 5);    // register (mi, r) is tile row lane_row0 + mi*32 + (r&3) + 8*(r>>2)
            const uint32_t row_base = (uint32_t)m0 + (uint32_t)lane_row0;
            if constexpr (DIRECT) {
                // round 0, thr = -inf: every score goes to slot (tile slot in round)*BM + row in tile of its query's buffer
#pragma unroll
                for (int ni = 0; ni < S::TN; ++ni) {
                    const int q = (int)n0 + acc.col(ni);
                    if (q < a.nq) {
                        uint64_t tslot_d; uint32_t tn_d; fast_divmod64((uint64_t)nat, (uint32_t)tn_count, tslot_d, tn_d);
                        uint64_t* cq = a.cand + (int64_t)q * a.cand_cap + (int64_t)tslot_d * S::BM + lane_row0;
#pragma unroll
                        for (int mi = 0; mi < S::TM; ++mi)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int ro = mi * 32 + (r & 3) + 8 * (r >> 2);
                                cq[ro] = make_key(acc.v[mi][ni][r], row_base + (uint32_t)ro);
                            }
                    }
                }
            } else {
#pragma unroll
                for (int ni = 0; ni < S::TN; ++ni) {
                    const uint32_t q = (uint32_t)n0 + (uint32_t)acc.col(ni);
                    const float t = thr_s[q];
#pragma unroll
                    for (int mi = 0; mi < S::TM; ++mi) {
                        // (a block-level pre-test — the max of the lane's 16 scores against the threshold, ONE branch per 32 x 32 block instead of 16
                        // taken ones — was measured 7 % SLOWER in an interleaved A/B on one device, profiles/r02/ab_coarse_blockmax_epilogue.txt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ro = mi * 32 + (r & 3) + 8 * (r >> 2);
                            const float s = acc.v[mi][ni][r];
                            const bool p = (s >= t);
                            const unsigned long long mask = __ballot(p);
                            if (__builtin_expect(mask != 0ull, 0)) {   // a survivor is rare (2.8 % of the registers in the final round): fall through without one
                                if (p) {
                                    const unsigned int slot = wcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                    u32x4 e = {__float_as_uint(s), row_base + (uint32_t)ro, q, 0u};
                                    __builtin_amdgcn_raw_buffer_store_b128(e, wlist, slot * 16u, 0, 0);
                                }
                                wcnt += (unsigned)__popcll(mask);
                            }
                        }
                    }
                }
            }
        };
    if constexpr (SMALLQ) gemm_nt_split<T, false>(a.xc, a.dpad, n_pad, a.qc, a.dpad, a.nq_pad, a.dpad, total, smem, coord, epi);
    else gemm_nt_pingpong<T, false, true>(a.xc, a.dpad, n_pad, a.qc, a.dpad, a.nq_pad, a.dpad, total, smem, coord, epi);   // corpus loads non-temporal
    if constexpr (!DIRECT) scatter_wave_lists(a, smem, wave_id, wcnt, ShapeC::NTHREADS);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Coarse scan for at most 32 queries (one question per hop of the KiRAG loop, interactive retrieval): a pure corpus STREAM.  The scan is HBM-bound
// (one MFMA per 1 KiB of corpus), so the kernel is built around bytes in flight, not around the matrix pipe:
//   * the queries' fragments for the WHOLE dimension live in registers (KT x 4 x 16 B per lane = 256 VGPRs at d = 1024; one wave per SIMD, the
//     unified 512-register file makes that possible), loaded once per launch: LDS holds nothing but corpus bytes;
//   * every wave streams its own 32-row tiles through a wave-private ring of RING K-tiles of 4 KiB (32 rows x 128 B, LDS-DMA, the same swizzled
//     image as the GEMM rings): no barrier anywhere in the loop, one counted s_waitcnt vmcnt per K-tile; 4 waves x 7 x 4 KiB = 112 KiB of corpus
//     bytes in flight per CU (48 KiB with the 128 x 128 producer / consumer tile);
//   * per 32 rows: KT x 4 MFMAs into ONE 32 x 32 accumulator, then the same threshold filter / direct store as k_coarse.
// Tile slots are 32 rows here (a.ntiles, a.tile_begin, a.tile_count, a.perm_mul are in units of 32 rows).
// ---------------------------------------------------------------------------------------------------------------------------------------------
constexpr int Q32_THREADS = 256;
template <int KT> struct Q32Ring { static constexpr int value = (KT % 8 == 0) ? 8 : (KT % 6 == 0) ? 6 : (KT % 4 == 0) ? 4 : 3; };   // (3: the int8 copy of 384-d rows)
template <int KT> constexpr int q32_lds() { return 4 * Q32Ring<KT>::value * 4096 > 2 * QBLK * 4 + 64 ? 4 * Q32Ring<KT>::value * 4096 : 2 * QBLK * 4 + 64; }

// (the body is a function with a __restrict__ corpus pointer on purpose: after inlining, the LDS-DMA carries that pointer's alias scope and the ring's
// ds_reads are marked as not aliasing it, which is what lets the compiler's waitcnt pass leave the counted vmcnt waits alone; without it it inserts
// s_waitcnt vmcnt(0) before every LDS read that follows an LDS-DMA, i.e. no K-tile would ever be in flight)
// MODE 0: threshold filter -> wave lists;  1: direct store of every score (round 0);  2: MARK — set bit `tile` of a.bitmap when any of the slot's
// 32 x 32 scores reaches its query's threshold (a.thr; the pre-scan of pass 2, see search_block)
// MODE 3: the BYTE pre-scan (byte_final_round).  The same ring over the int8 copy: a.xc / a.qc point at bytes, a.dpad counts 2-byte units, so a K-tile is
// 128 int8 per row and every address below is unchanged; the queries come as two byte planes (a.qc hi, a.qc2 lo: q ~ sq (hi + lo / 254)), two
// v_mfma_i32_32x32x32_i8 chains give the exact integer dot products Ia, Ib, and the row's bit in a.bitmap is set when sx8[row] (254 Ia + Ib) >= a.thr[query]
// for some query.
// (Integer sums are order-free, so all that matters of the MFMA's k layout is that A and B use the same one: both read 16-byte chunk 2 ks + (lane >> 5).)
template <class T, int MODE, int KT>
__device__ __forceinline__ void coarse_q32_body(const CoarseArgs& a, const uint16_t* __restrict__ xc, const uint16_t* __restrict__ qc, char* smem) {
    constexpr int RING = Q32Ring<KT>::value;
    static_assert(KT % RING == 0, "ring slots must be static in the unrolled K loop");
    const int lane = threadIdx.x & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* ring = smem + wave_id * (RING * 4096);
    const __amdgpu_buffer_rsrc_t wlist = __builtin_amdgcn_make_buffer_rsrc(
        a.blk_list + ((int64_t)blockIdx.x * ShapeC::NWAVE + wave_id) * WLISTCAP, 0, WLISTCAP * 16, 0x00020000);
    unsigned int wcnt = 0;
    // query fragments (B operand of mfma(a, b)): lane -> query lane & 31, k = 16 ks + 8 (lane >> 5) .. + 7
    uint4 bq[KT * 4];
    {
        const uint16_t* qrow = qc + (int64_t)(lane & 31) * a.dpad + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KT * 4; ++ks) bq[ks] = *reinterpret_cast<const uint4*>(qrow + ks * 16);
    }
    uint4 bq2[MODE == 3 ? KT * 4 : 1];
    if constexpr (MODE == 3) {
        const uint16_t* qrow = a.qc2 + (int64_t)(lane & 31) * a.dpad + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < KT * 4; ++ks) bq2[ks] = *reinterpret_cast<const uint4*>(qrow + ks * 16);
    }
    constexpr bool DIRECT = MODE == 1;
    const float thr = DIRECT ? 0.f : a.thr[lane & 31];
    const uint32_t q = (uint32_t)(lane & 31);
    // this wave's tile slots: s = w, w + W, w + 2W, ... of the round's [0, tile_count)
    const int64_t W = (int64_t)gridDim.x * 4, w0 = (int64_t)blockIdx.x * 4 + wave_id;
    const int64_t my = a.tile_count > w0 ? (a.tile_count - w0 + W - 1) / W : 0;
    if (my > 0) {
        auto tile_row0 = [&](int64_t i) -> int64_t {          // first corpus row of this wave's i-th tile (i clamped: dummy re-loads past the end)
            const int64_t slot = w0 + (i < my ? i : my - 1) * W;
            uint64_t qq; uint32_t tile;
            fast_divmod64((uint64_t)(a.tile_begin + slot) * (uint64_t)a.perm_mul, (uint32_t)a.ntiles, qq, tile);
            return (int64_t)tile * 32;
        };
        // DMA cursor: (tile index, K-tile); per-lane source: piece p = rows 8p .. 8p+7, lane -> row 8p + (lane >> 3), 16-B chunk (lane & 7) ^ swizzle
        const int r8 = lane >> 3;
        int64_t pf_tile = 0; int pf_kt = 0;
        const char* pf_base = reinterpret_cast<const char*>(xc + tile_row0(0) * a.dpad);
        uint32_t src_off[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = 8 * p + r8;
            src_off[p] = (uint32_t)(row * a.dpad * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
        }
        auto stage = [&](int slot) {
            char* dst = ring + slot * 4096;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                __builtin_amdgcn_global_load_lds((gbl_void*)(pf_base + pf_kt * 128 + src_off[p]), (lds_void*)(dst + p * 1024), 16, 0, 2);   // nt: once-through stream
            if (++pf_kt == KT) { pf_kt = 0; ++pf_tile; pf_base = reinterpret_cast<const char*>(xc + tile_row0(pf_tile) * a.dpad); }
        };
#pragma unroll
        for (int p = 0; p < RING - 1; ++p) stage(p);
        const int frow = lane & 31, fh = lane >> 5, fswz = (frow >> 1) & 7;
        const int lane_row0 = 4 * fh;
        for (int64_t i = 0; i < my; ++i) {
            const int64_t m0 = tile_row0(i);
            f32x16 acc;
            i32x16 ia, ib;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; ia[r] = 0; ib[r] = 0; }
            float4 sc4[4];                                    // MODE 3: scales of this lane's 16 rows (4 fh + 8 j .. + 3), requested ahead of the K loop
            if constexpr (MODE == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) sc4[j] = *reinterpret_cast<const float4*>(a.sx8 + m0 + 8 * j + 4 * (lane >> 5));
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                // the slot of the PREVIOUS K-tile is free (its fragments were consumed by MFMAs that are already issued): refill it first, then wait
                // for this K-tile: RING-1 younger ones stay in flight
                stage((kt + RING - 1) % RING);
                wait_vmcnt<4 * (RING - 1)>();
                const char* st = ring + (kt % RING) * 4096 + frow * 128;
                uint4 af[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) af[ks] = *reinterpret_cast<const uint4*>(st + (((2 * ks + fh) ^ fswz) << 4));
                if constexpr (MODE == 3) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        ia = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, af[ks]), __builtin_bit_cast(i32x4, bq[kt * 4 + ks]), ia, 0, 0, 0);
                        ib = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, af[ks]), __builtin_bit_cast(i32x4, bq2[kt * 4 + ks]), ib, 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) acc = T::mfma(af[ks], bq[kt * 4 + ks], acc);
                }
                asm volatile("" ::: "memory");                // keep the next refill behind these LDS reads in program order
            }
            const uint32_t row_base = (uint32_t)m0 + (uint32_t)lane_row0;
            if constexpr (DIRECT) {
                if (q < (uint32_t)a.nq) {
                    uint64_t* cq = a.cand + (int64_t)q * a.cand_cap + (w0 + i * W) * 32 + lane_row0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        cq[ro] = make_key(acc[r], row_base + (uint32_t)ro);
                    }
                }
            } else if constexpr (MODE == 2) {
                // register r holds, for the lane's query, row (r & 3) + 8 (r >> 2) of the tile in lanes 0..31 and that row + 4 in lanes 32..63;
                // NaN scores (padding rows, NaN rows) never mark.  The 32 rows of a tile share one bitmap word (m0 is a multiple of 32).
                unsigned int bits = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned long long mask = __ballot(acc[r] >= thr);
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if ((unsigned)mask) bits |= 1u << ro;
                    if ((unsigned)(mask >> 32)) bits |= 1u << (ro + 4);
                }
                if (bits && lane == 0) atomicOr(a.bitmap + (m0 >> 5), bits);
            } else if constexpr (MODE == 3) {
                // |Ia| <= 127 * 127 * 1024 < 2^24: exact in fp32; the fma rounds once (covered by the slack in k_scan8_prep).  Padding rows carry a NaN scale.
                unsigned int bits = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sr = (r & 3) == 0 ? sc4[r >> 2].x : (r & 3) == 1 ? sc4[r >> 2].y : (r & 3) == 2 ? sc4[r >> 2].z : sc4[r >> 2].w;
                    const float sc = sr * fmaf(254.f, (float)ia[r], (float)ib[r]);
                    const unsigned long long mask = __ballot(sc >= thr);
                    const int ro = (r & 3) + 8 * (r >> 2);
                    const unsigned lo = (unsigned)mask, hi = (unsigned)(mask >> 32);      // the queries (lanes 0..31) that want row ro / row ro + 4
                    if (lo) { bits |= 1u << ro; if (lane == 0) atomicOr(a.qmask + m0 + ro, lo); }
                    if (hi) { bits |= 1u << (ro + 4); if (lane == 0) atomicOr(a.qmask + m0 + ro + 4, hi); }
                }
                // fire-and-forget (no returned value: an atomic that returns one would be waited for with vmcnt, i.e. behind the whole ring of K-tiles in
                // flight — measured: appending the rows to a list from here cost 0.1 ms per 5M-row scan)
                if (bits && lane == 0) atomicOr(a.bitmap + (m0 >> 5), bits);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    const float sc = acc[r];
                    const bool p = (sc >= thr);
                    const unsigned long long mask = __ballot(p);
                    if (mask) {
                        if (p) {
                            const unsigned int slot = wcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                            u32x4 e = {__float_as_uint(sc), row_base + (uint32_t)ro, q, 0u};
                            __builtin_amdgcn_raw_buffer_store_b128(e, wlist, slot * 16u, 0, 0);
                        }
                        wcnt += (unsigned)__popcll(mask);
                    }
                }
            }
        }
    }
    if constexpr (MODE == 0) scatter_wave_lists(a, smem, wave_id, wcnt, Q32_THREADS);
}

template <class T, int MODE, int KT>
__global__ __launch_bounds__(Q32_THREADS, 1) void k_coarse_q32(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    coarse_q32_body<T, MODE, KT>(a, a.xc, a.qc, smem);
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// High-precision pass for the queries the 16-bit scan could not certify (dense clusters below the bf16 / f16 resolution, candidate-buffer
// overflows): the SAME filter / select / certified re-rank pipeline, but the scan multiplies the fp32 MASTER rows by the fp32 queries with
// fp64 products and fp64 accumulation (v_mfma_f64_16x16x4_f64: products of two fp32 values are exact in fp64), so its error bound is ~1e-11
// instead of ~2.5e-3: the certificate then fails only on more than RERANK_MAX rows within a few fp32 ulps of the k-th score.  All flagged queries
// of a group (up to FINE_QMAX = 32 at d <= 1024) share ONE pass over the corpus: 4 KiB per row once per group, instead of once per flagged query
// as the exact scan does (round 1: 20 GB per flagged query at 5M rows).
//   * the group's queries sit in LDS as fp32 (row stride dk * 4 + 16 B: the 16 queries of a B fragment fall on different banks);
//   * every wave streams its own 32-row slots (two 16-row MFMA tiles) straight from HBM to registers, one 128-element chunk (16 float4 per lane)
//     in flight behind the chunk being multiplied; A fragment: lane -> row l & 15, k = 32 j + 8 (l >> 4) + c; B fragment: query l & 15, same k;
//     the k order inside the sum is irrelevant as long as A and B agree (step (j, c) pairs component c of both float4 pairs);
//   * epilogue: D[row (l >> 4) + 4 r][query l & 15], fp64 -> fp32, then the same direct store / threshold filter as the 16-bit scans.
// Tile slots are 32 rows (a.ntiles, a.tile_begin, a.tile_count, a.perm_mul in units of 32 rows), a.nq <= 16 NT queries.
// ---------------------------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) double f64x4;
constexpr int FINE_THREADS = 512;
constexpr int FINE_QMAX = 32;
constexpr int FINE_DMAX = 2048;                       // 16 queries x 2048 x 4 B = 128 KiB of LDS
static inline int fine_lds_bytes(int d, int gq) { const int dk = (int)round_up(d, 128); const int b = gq * (dk * 4 + 16) + 64; return b > 3 * QBLK * 4 ? b : 3 * QBLK * 4; }

template <bool DIRECT, int NT>
__global__ __launch_bounds__(FINE_THREADS, 2) void k_fine(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int d = a.d;
    const int dk = (d + 127) / 128 * 128;              // zero-filled tail: chunks of 128 elements
    const int qstride = dk * 4 + 16;
    for (int idx = threadIdx.x; idx < 16 * NT * (dk / 4); idx += FINE_THREADS) {
        const int g = idx / (dk / 4), c4 = idx - g * (dk / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g < a.nq && c4 * 4 < d) v = *reinterpret_cast<const float4*>(a.qf + (int64_t)g * d + c4 * 4);
        *reinterpret_cast<float4*>(smem + g * qstride + c4 * 16) = v;
    }
    float thr[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) thr[nt] = DIRECT ? 0.f : a.thr[16 * nt + (lane & 15)];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t wlist = __builtin_amdgcn_make_buffer_rsrc(
        a.blk_list + ((int64_t)blockIdx.x * ShapeC::NWAVE + wave_id) * WLISTCAP, 0, WLISTCAP * 16, 0x00020000);
    unsigned int wcnt = 0;
    const int64_t W = (int64_t)gridDim.x * 8, w0 = (int64_t)blockIdx.x * 8 + wave_id;
    const int64_t my = a.tile_count > w0 ? (a.tile_count - w0 + W - 1) / W : 0;
    const int kq = 8 * (lane >> 4);                    // this lane's k offset inside a 32-element step
    const char* qlds = smem + (lane & 15) * qstride + kq * 4;
    const int nchunk = dk / 128;
    for (int64_t i = 0; i < my; ++i) {
        uint64_t qq; uint32_t tile;
        fast_divmod64((uint64_t)(a.tile_begin + w0 + i * W) * (uint64_t)a.perm_mul, (uint32_t)a.ntiles, qq, tile);
        const int64_t m0 = (int64_t)tile * 32;
        const int64_t n_eff = a.rowlist ? a.nlist : a.n;   // with a row list, "row" below is a position in the list; the stored key carries the real row
        const float* pa[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int64_t row = m0 + 16 * t + (lane & 15);
            if (row >= n_eff) row = n_eff - 1;         // clamped rows are masked in the epilogue
            if (a.rowlist) row = (int64_t)a.rowlist[row];
            pa[t] = a.xf + row * d + kq;
        }
        f64x4 acc[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[t][nt] = f64x4{0.0, 0.0, 0.0, 0.0};
        float4 cur[4][2][2], nxt[4][2][2];
        auto load_chunk = [&](float4 (&buf)[4][2][2], int c) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int kk = c * 128 + u * 32 + kq + 4 * h;
                        buf[u][t][h] = (kk < d) ? *reinterpret_cast<const float4*>(pa[t] + c * 128 + u * 32 + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
        };
        load_chunk(cur, 0);
        for (int c = 0; c < nchunk; ++c) {
            if (c + 1 < nchunk) load_chunk(nxt, c + 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 bq[NT][2];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        bq[nt][h] = *reinterpret_cast<const float4*>(qlds + nt * 16 * qstride + (c * 128 + u * 32 + 4 * h) * 4);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float av[2][4] = {{cur[u][0][h].x, cur[u][0][h].y, cur[u][0][h].z, cur[u][0][h].w},
                                            {cur[u][1][h].x, cur[u][1][h].y, cur[u][1][h].z, cur[u][1][h].w}};
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const float bvv[4] = {bq[nt][h].x, bq[nt][h].y, bq[nt][h].z, bq[nt][h].w};
                            const double b = (double)bvv[cc];
#pragma unroll
                            for (int t = 0; t < 2; ++t)
                                acc[t][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[t][cc], b, acc[t][nt], 0, 0, 0);
                        }
                    }
                }
            }
            if (c + 1 < nchunk) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int h = 0; h < 2; ++h) cur[u][t][h] = nxt[u][t][h];
            }
        }
        // epilogue: register r of tile t, n-tile nt = row m0 + 16 t + (lane >> 4) + 4 r, query 16 nt + (lane & 15)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const uint32_t q = (uint32_t)(16 * nt + (lane & 15));
            if constexpr (DIRECT) {
                if (q < (uint32_t)a.nq) {
                    uint64_t* cq = a.cand + (int64_t)q * a.cand_cap + (w0 + i * W) * 32;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ro = 16 * t + (lane >> 4) + 4 * r;
                            const int64_t row = m0 + ro;
                            cq[ro] = row < n_eff ? make_key((float)acc[t][nt][r], a.rowlist ? a.rowlist[row] : (uint32_t)row) : 0ull;
                        }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t row = m0 + 16 * t + (lane >> 4) + 4 * r;
                        const float sc = (float)acc[t][nt][r];
                        const bool p = (sc >= thr[nt]) && row < n_eff;        // thr = +inf for padded queries
                        const unsigned long long mask = __ballot(p);
                        if (mask) {
                            if (p) {
                                const unsigned int slot = wcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                                u32x4 e = {__float_as_uint(sc), a.rowlist ? a.rowlist[row] : (uint32_t)row, q, 0u};
                                __builtin_amdgcn_raw_buffer_store_b128(e, wlist, slot * 16u, 0, 0);
                            }
                            wcnt += (unsigned)__popcll(mask);
                        }
                    }
            }
        }
    }
    if constexpr (!DIRECT) scatter_wave_lists(a, smem, wave_id, wcnt, FINE_THREADS);
}

// compacted copy of the flagged queries: dst[g] = src[qidx[g]]
__global__ void k_gather_rows(const float* __restrict__ src, const int* __restrict__ qidx, float* __restrict__ dst, int d) {
    const int g = blockIdx.x;
    const float* s = src + (int64_t)qidx[g] * d;
    for (int i = threadIdx.x * 4; i < d; i += blockDim.x * 4) *reinterpret_cast<float4*>(dst + (int64_t)g * d + i) = *reinterpret_cast<const float4*>(s + i);
}

// bitmap (one bit per row) -> unordered list of the marked rows; *count += its length (the order is irrelevant: the re-rank sorts by (score, row)).
// One reservation per block of 1024 words (round 5: one per wave serialised ~1500 atomics on one address, 15 us of the 22 this kernel took at 5M rows).
// CLEAN: every word read is written back as zero, so the bitmap is all-zero again for the next scan without a memset in between.
template <bool CLEAN>
__global__ __launch_bounds__(1024) void k_compact_rows(uint32_t* __restrict__ bitmap, int64_t nwords, uint32_t* __restrict__ rowlist, unsigned int* __restrict__ count) {
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int bbase;
    const int64_t w = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    uint32_t bits = w < nwords ? bitmap[w] : 0u;
    if (CLEAN && bits) bitmap[w] = 0u;
    const unsigned c = __popc(bits);
    unsigned incl = c;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) { const unsigned t = __shfl_up(incl, dlt, 64); if ((int)(threadIdx.x & 63) >= dlt) incl += t; }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wsum[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tot = 0u;
        for (int i = 0; i < 16; ++i) { const unsigned t = wsum[i]; wsum[i] = tot; tot += t; }
        bbase = tot ? atomicAdd(count, tot) : 0u;
    }
    __syncthreads();
    unsigned pos = bbase + wsum[wave] + incl - c;
    while (bits) { const int b = __ffs(bits) - 1; bits &= bits - 1; rowlist[pos++] = (uint32_t)(w * 32 + b); }
}

// thr_mark[g] = theta1[qidx[g]] for g < n, +inf for the padding up to 32 (the marking scan's per-query thresholds)
__global__ void k_gather_theta(const float* __restrict__ theta1, const int* __restrict__ qidx, int n, float* __restrict__ out) {
    const int g = threadIdx.x;
    if (g < 32) out[g] = g < n ? theta1[qidx[g]] : INFINITY;
}

// state reset + error bound of the high-precision pass for queries [0, nq) of the compacted list (blocks nq .. nq_pad-1: padding, thr = +inf).
//   |canonical - stored fine score| <= |fine64 - exact| + two fp32 roundings
//     fine64: fp64 accumulation of exact products in hardware order: <= 4 (d + 8) 2^-53 |q|.|x|  (a factor 4 over the round-to-nearest bound:
//     the MFMA's internal rounding is not documented)
template <int DUMMY = 0>
__global__ __launch_bounds__(64) void k_prep_fine(const float* __restrict__ qf, int nq, int d, const float* __restrict__ bounds, float* __restrict__ eps,
                                                  float* __restrict__ thr, uint32_t* __restrict__ cnt, uint32_t* __restrict__ flags) {
    const int q = blockIdx.x, lane = threadIdx.x;
    float q2 = 0.f;
    if (q < nq)
        for (int i = lane * 4; i < d; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(qf + (int64_t)q * d + i);
            q2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) q2 += __shfl_xor(q2, m, 64);
    if (lane == 0) {
        const float xn = bounds[0] + bounds[1];          // max |x| <= max |c(x)| + max |x - c(x)|
        const float qn = sqrtf(q2) * 1.0001f;
        float e = (2.4e-7f + 4.5e-16f * (float)(d + 8)) * qn * xn;
        e = e * 1.001f + 1e-37f;
        eps[q] = (q < nq) ? e : 0.f;
        thr[q] = (q < nq) ? -INFINITY : INFINITY;
        cnt[q] = 0u;
        flags[q] = 0u;
    }
}

// rows [n, round_up(n, 256)) of the 16-bit copy <- NaN
__global__ void k_pad_nan(uint16_t* __restrict__ xc, int64_t n, int64_t n_pad, int dpad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (n_pad - n) * dpad) xc[n * dpad + i] = 0x7FC0;   // NaN in bf16 and in f16 (0x7FC0 = f16 NaN too: exponent 11111, mantissa != 0)
}

// ---------------------------------------------------------------------------------------------------------
// selection by value (no sorting of the candidate buffers): MSB-first 8-bit radix select over the ordered score bits
// ---------------------------------------------------------------------------------------------------------
// R-th largest (1-based) of hi32(s[0..m)), for a block of NT threads; hist: 256 + 4 words of LDS.  Requires 1 <= R <= m.
template <int NT>
__device__ __forceinline__ uint32_t radix_select_desc(const uint64_t* __restrict__ s, int m, int R, unsigned int* hist, int tid) {
    uint32_t prefix = 0u, pmask = 0u;
    int remaining = R;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = tid; i < 256; i += NT) hist[i] = 0u;
        __syncthreads();
        for (int i = tid; i < m; i += NT) {
            const uint32_t hi = (uint32_t)(s[i] >> 32);
            if ((hi & pmask) == prefix) atomicAdd(&hist[(hi >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {   // wave 0: lane l owns bins 4l .. 4l+3; suffix sums from the top bin down
            const unsigned h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const unsigned own = h0 + h1 + h2 + h3;
            unsigned incl = own;   // inclusive suffix sum over lanes >= tid
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_down(incl, d, 64); if (tid + d < 64) incl += t; }
            const unsigned above = incl - own;                       // keys in bins of higher lanes
            if (above < (unsigned)remaining && incl >= (unsigned)remaining) {   // the R-th key falls into this lane's bins
                unsigned c = above; int bin = 4 * tid + 3;
                if (c + h3 >= (unsigned)remaining) bin = 4 * tid + 3;
                else { c += h3; if (c + h2 >= (unsigned)remaining) bin = 4 * tid + 2;
                       else { c += h2; if (c + h1 >= (unsigned)remaining) bin = 4 * tid + 1; else { c += h1; bin = 4 * tid; } } }
                hist[256] = (unsigned)bin; hist[257] = c;
            }
        }
        __syncthreads();
        prefix |= hist[256] << shift; pmask |= 255u << shift;
        remaining -= (int)hist[257];
        __syncthreads();
    }
    return prefix;
}

// between rounds: keep the entries with score >= (keep-th best score), publish thr[q] = thr_rank-th best score (thr_rank <= keep).
// `preset` > 0: the buffer was filled by the direct round with `preset` slots (cnt unused).
template <int NT>   // threads per block: 256 for blocks of many queries, 1024 for the <= 32-query stream (one block per query: the few blocks are a latency chain)
__global__ __launch_bounds__(NT) void k_select(uint64_t* __restrict__ cand, int cand_cap, uint32_t* __restrict__ cnt,
                                                uint32_t* __restrict__ flags, float* __restrict__ thr, int keep, int thr_rank, int preset) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem);
    unsigned int* hist = reinterpret_cast<unsigned int*>(smem + (size_t)cand_cap * sizeof(uint64_t));   // 256 + 4
    unsigned int& outc = hist[259];
    const int q = blockIdx.x, tid = threadIdx.x;
    uint32_t m = preset > 0 ? (uint32_t)preset : cnt[q];
    if (m > (uint32_t)cand_cap) {   // entries beyond the capacity were dropped: the query goes to the exact scan
        if (tid == 0) flags[q] |= 1u;
        m = (uint32_t)cand_cap;
    }
    if (m == 0) { if (tid == 0) cnt[q] = 0; return; }
    uint64_t* c = cand + (int64_t)q * cand_cap;
    for (int i = tid; i < (int)m; i += NT) s[i] = c[i];
    if (tid == 0) outc = 0u;
    __syncthreads();
    uint32_t vkeep = 0u;
    if ((int)m >= thr_rank) {
        const uint32_t vthr = radix_select_desc<NT>(s, (int)m, thr_rank, hist, tid);
        if (tid == 0) thr[q] = ord_f32(vthr);      // NaN (vthr == 0) compares false everywhere: nothing more is appended, the query ends up flagged
        vkeep = vthr;
    }
    if ((int)m > keep) vkeep = (keep == thr_rank && (int)m >= thr_rank) ? vkeep : radix_select_desc<NT>(s, (int)m, keep, hist, tid);
    else vkeep = 0u;
    // compact: everything with score >= the keep-th best (ties kept; zero keys = padding dropped)
    for (int i = tid; i < (int)m; i += NT) {
        const uint64_t key = s[i];
        if (key != 0ull && (uint32_t)(key >> 32) >= vkeep) c[atomicAdd(&outc, 1u)] = key;
    }
    __syncthreads();
    if (tid == 0) cnt[q] = outc;
}

// exactness certificate + fp64 re-rank of the candidates with coarse score >= b_k - 2 eps   (buffer order is arbitrary)
constexpr int MERGE_MAX = 8192;   // nshards * k entries per query (8 shards x k = 1024): 96 KiB of LDS in the device merge, 64 KiB in k_global_theta
constexpr int RERANK_MAX = 2048;
template <int NCH>   // NCH 256-element steps cover a row: d <= 256 NCH
__global__ __launch_bounds__(256) void k_rerank(const uint64_t* __restrict__ cand, int cand_cap, const uint32_t* __restrict__ cnt,
                                                uint32_t* __restrict__ flags, const float* __restrict__ thr, const float* __restrict__ eps,
                                                const float* __restrict__ qf, const float* __restrict__ xf, int d, int k, int preset, int rmax,
                                                float* __restrict__ out_s, int64_t* __restrict__ out_r, uint32_t* __restrict__ nrer,
                                                const int* __restrict__ qmap, int force_exact, float* __restrict__ theta_out,
                                                const float* __restrict__ theta_ext = nullptr) {
    // theta_ext[q] (row-sharded search, exchange BEFORE the re-rank: kr_index_search_rerank_async): a bound derived from the k-th best coarse score of ALL
    // shards; rows of this shard below it cannot be in the GLOBAL top-k, so the re-rank looks at max(theta, theta_ext) and may return FEWER than k rows
    // (the tail is padded with (-inf, -1): the merge of the shards' lists treats id < 0 as padding)
    // theta_out[q] (pass 1 only): b_k - 2 eps of THIS pass = a bound below which a row's score in this pass's arithmetic rules it out of the top-k
    // (-inf when the query has fewer than k candidates); pass 2 pre-scans with it so that its fp64 work touches only slots that can matter
    // qmap != nullptr: the blocks work on a compacted query list (the high-precision pass over flagged queries); results go to row qmap[q]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned long long exact_limbs[4][EXACT_NLIMB];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem);                                    // [cand_cap] copy of the buffer
    uint64_t* sel = s + cand_cap;                                                       // [rmax]
    unsigned int* hist = reinterpret_cast<unsigned int*>(sel + rmax);                   // 256 + 4
    unsigned int& rc = hist[259];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int m = preset > 0 ? preset : (int)cnt[q];
    bool ok = (flags[q] == 0u);
    if (m > cand_cap) { m = cand_cap; ok = false; }
    if (m < k) {   // fewer candidates than k (NaN rows, overflow truncation): the exact scan decides
        if (tid == 0) { flags[q] |= 2u; nrer[q] = 0; if (theta_out) theta_out[q] = -INFINITY; }
        return;
    }
    const uint64_t* c = cand + (int64_t)q * cand_cap;
    for (int i = tid; i < m; i += 256) s[i] = c[i];
    if (tid == 0) rc = 0u;
    __syncthreads();
    const float bk = ord_f32(radix_select_desc<256>(s, m, k, hist, tid));               // k-th best coarse score (NaN if fewer than k real scores)
    float theta = bk - 2.f * eps[q];
    if (theta_out && tid == 0) theta_out[q] = (theta == theta) ? theta : -INFINITY;        // the LOCAL bound: pass 2 answers with the shard's own top-k
    if (theta_ext) { const float te = theta_ext[q]; if (te > theta) theta = te; }
    // the buffer is complete for coarse scores >= thr (everything at or above the threshold of the last round was appended / kept)
    const bool certified = ok && (theta > thr[q]);
    for (int i = tid; i < m; i += 256) {
        const uint64_t key = s[i];
        if (key_score(key) >= theta) { const unsigned p = atomicAdd(&rc, 1u); if (p < (unsigned)rmax) sel[p] = key; }
    }
    __syncthreads();
    const int r_all = (int)rc;
    const int r = r_all < rmax ? r_all : rmax;
    if (tid == 0) { nrer[q] = (uint32_t)r; if (!certified || r_all > rmax || (r_all < k && !theta_ext)) flags[q] |= 2u; }
    int P = 1; while (P < r) P <<= 1;
    if (P < 1) P = 1;
    __syncthreads();
    const float* qv = qf + (int64_t)q * d;
    {
        float4 qr[NCH];                                        // the query stays in registers for all rows of this wave
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int i = lane * 4 + j * 256;
            qr[j] = (i < d) ? *reinterpret_cast<const float4*>(qv + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // two candidate rows per wave and step (i and i + 4); the rows of the NEXT step are requested before this step's arithmetic, so a wave pays
        // the ~2 us of a scattered 4-KiB gather once, not once per pair (this loop was a latency chain: 275 rows per query = 35 round trips)
        RowPair<NCH> cur, nxt;
        int i = wave;
        bool v0 = i < r, v1 = i + 4 < r;
        uint32_t row0 = v0 ? key_row(sel[i]) : 0u, row1 = v1 ? key_row(sel[i + 4]) : row0;
        if (v0) load_rows2<NCH>(cur, xf + (int64_t)row0 * d, xf + (int64_t)row1 * d, d, lane);
        for (; i < P; i += 8) {
            const int in = i + 8;
            const bool nv0 = in < r, nv1 = in + 4 < r;
            const uint32_t nrow0 = nv0 ? key_row(sel[in]) : 0u, nrow1 = nv1 ? key_row(sel[in + 4]) : nrow0;
            if (nv0) load_rows2<NCH>(nxt, xf + (int64_t)nrow0 * d, xf + (int64_t)nrow1 * d, d, lane);
            float e0 = 0.f, e1 = 0.f;
            if (v0) score_rows2<NCH>(qr, qv, cur, xf + (int64_t)row0 * d, xf + (int64_t)row1 * d, d, lane, exact_limbs[wave], force_exact != 0, e0, e1);
            if (lane == 0) {
                sel[i] = v0 ? make_key(e0, row0) : 0ull;
                if (i + 4 < P) sel[i + 4] = v1 ? make_key(e1, row1) : 0ull;
            }
            cur = nxt; v0 = nv0; v1 = nv1; row0 = nrow0; row1 = nrow1;
        }
    }
    bitonic_sort_desc(sel, P, tid, 256);
    const int64_t qo = qmap ? (int64_t)qmap[q] : (int64_t)q;
    for (int j = tid; j < k; j += 256) {
        if (theta_ext && j >= r) { out_s[qo * k + j] = -INFINITY; out_r[qo * k + j] = -1; }      // fewer than k rows of this shard above the global bound
        else if (j < P) { out_s[qo * k + j] = key_score(sel[j]); out_r[qo * k + j] = (int64_t)key_row(sel[j]); }
    }
}

// Row-sharded search, exchange before the re-rank.  k_local_topk: the k best COARSE scores of this shard's candidate buffer per query (unsorted; -inf when the
// buffer holds fewer than k) followed by the query's error bound eps — k + 1 floats per query, what a shard contributes to the all-gather.
__global__ __launch_bounds__(256) void k_local_topk(const uint64_t* __restrict__ cand, int cand_cap, const uint32_t* __restrict__ cnt, const float* __restrict__ eps,
                                                    int k, int preset, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem);
    unsigned int* hist = reinterpret_cast<unsigned int*>(s + cand_cap);                 // 256 + 4
    unsigned int& wc = hist[259];
    const int q = blockIdx.x, tid = threadIdx.x;
    float* o = out + (int64_t)q * (k + 1);
    int m = preset > 0 ? preset : (int)cnt[q];
    if (m > cand_cap) m = cand_cap;
    if (tid == 0) o[k] = eps[q];
    if (m < k) { for (int j = tid; j < k; j += 256) o[j] = -INFINITY; return; }
    const uint64_t* c = cand + (int64_t)q * cand_cap;
    for (int i = tid; i < m; i += 256) s[i] = c[i];
    if (tid == 0) wc = 0u;
    __syncthreads();
    const uint32_t vk = radix_select_desc<256>(s, m, k, hist, tid);                      // k-th best (order-preserving bits)
    for (int i = tid; i < m; i += 256) {
        const uint32_t v = (uint32_t)(s[i] >> 32);
        if (v > vk) { const unsigned pos = atomicAdd(&wc, 1u); o[pos] = ord_f32(v); }    // strictly better than the k-th: fewer than k of them
    }
    __syncthreads();
    for (int j = (int)wc + tid; j < k; j += 256) o[j] = ord_f32(vk);                    // the k-th best and its ties fill the rest
}

// k_global_theta: the gathered [nshards][nq][k + 1] lists -> per query the k-th best coarse score of ALL shards b_G and the largest error bound e_max;
// theta_ext[q] = b_G - (e_max + eps_local[q]): the k rows with the best coarse scores have exact scores >= b_G - e_max, so a row of the exact global top-k
// has exact >= b_G - e_max and therefore, on a shard with bound eps_local, coarse >= b_G - e_max - eps_local.
__global__ __launch_bounds__(256) void k_global_theta(const float* __restrict__ gathered, int nshards, int64_t shard_stride, int k, const float* __restrict__ eps_local,
                                                      float* __restrict__ theta_ext) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem);
    unsigned int* hist = reinterpret_cast<unsigned int*>(s + (size_t)nshards * k);
    const int q = blockIdx.x, tid = threadIdx.x;
    const int m = nshards * k;
    for (int i = tid; i < m; i += 256) {
        const int w = i / k, j = i - w * k;
        s[i] = make_key(gathered[w * shard_stride + (int64_t)q * (k + 1) + j], (uint32_t)i);
    }
    __syncthreads();
    const float bg = ord_f32(radix_select_desc<256>(s, m, k, hist, tid));
    if (tid == 0) {
        float emax = 0.f;
        for (int w = 0; w < nshards; ++w) emax = fmaxf(emax, gathered[w * shard_stride + (int64_t)q * (k + 1) + k]);
        const float t = bg - (emax + eps_local[q]) * 1.000001f;
        theta_ext[q] = (t == t) ? t : -INFINITY;
    }
}

// ---------------------------------------------------------------------------------------------------------
// byte pre-scan of small query blocks (<= BYTE_NQ_MAX queries: the KiRAG loop's 1-2 queries per hop, one question at a time)
// ---------------------------------------------------------------------------------------------------------
// A block of up to 32 queries is an HBM-bound stream over the 16-bit copy (2 dpad bytes per row).  The final round of that scan — 94 % of the rows — only has
// to find the rows that can still be in a query's top-k, and half the bytes suffice for that: an int8 copy with one scale per row ((x - mu) w ~ sx8 x8: k_mu_final) is streamed
// instead, every row whose byte score reaches
//     theta8[q] = (k-th best 16-bit score of the rows seen in the earlier rounds) - eps16[q] - eps8[q]
// for some query is marked, and only the marked rows (a few hundred to a few thousand per query) get their 16-bit scores (k_score_list) and enter the
// candidate buffers exactly as the final 16-bit round would have entered them.  Exactness: the k rows with the best 16-bit scores seen so far have exact
// scores >= kth16 - eps16, so a row of the final top-k has exact >= kth16 - eps16 and byte score >= kth16 - eps16 - eps8; an unmarked row is strictly below
// the k-th exact score of rows already seen and can never be returned.  k_rerank's certificate is untouched: rows missing from the buffer are either below
// the round threshold (as before) or unmarked (ruled out here).
//   q.x = q.mu + u.r with u = q / w, r = (x - mu) w;  |u.r - u^.r^| <= |u| |r - r^| + |u - u^| |r^|,  r^ = sx8 x8 (bounds8[0] = max |r - r^|),
//   u^ = sq (qa + qb / 254)  (two planes: |u - u^| ~ 1e-5 |u|);  eps8 = that + the fp32 roundings involved (k_scan8_prep)
// The int8 copy is derived data (built lazily, ensure_byte_copy); an index with a non-finite element anywhere never takes this path (bounds8 word 1).

// centre and axis weights of the int8 copy, from rows [0, m): mu = their mean, w_i = the power of two nearest to sigma_rms / sigma_i (clamped to
// [1/16, 16]); a row is stored as x8 ~ (x - mu) * w / sx8 and a query enters as q / w, so that q.(x - mu) = (q / w).((x - mu) w) exactly (powers of
// two: no rounding).  ANY centre and ANY positive weights are valid — they only decide how tight the bound is: the mean removes what all embeddings of
// one encoder share (e5 / bge rows have a common direction), the weights keep a few large-variance axes ("rogue dimensions") from setting every row's
// scale.  Two deterministic stages: MU_PARTS partial sums, then one block.
constexpr int CNT_STRIDE = 1024;                     // words between the spread candidate counters (see k_score_list)
constexpr int MU_PARTS = 256;
constexpr int64_t MU_ROWS = 65536;
__global__ __launch_bounds__(256) void k_mu_partial(const float* __restrict__ xf, int64_t m, int d, int dpad8, float* __restrict__ part) {
    const int64_t per = (m + MU_PARTS - 1) / MU_PARTS, r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < m ? r0 + per : m;
    for (int i = threadIdx.x; i < dpad8; i += 256) {
        float acc = 0.f, acc2 = 0.f;
        if (i < d) for (int64_t r = r0; r < r1; ++r) { const float v = xf[r * d + i]; acc += v; acc2 += v * v; }
        part[((int64_t)blockIdx.x * 2) * dpad8 + i] = acc;
        part[((int64_t)blockIdx.x * 2 + 1) * dpad8 + i] = acc2;
    }
}
// one block of 1024 threads (dpad8 <= 1024): mu8 = [mu | w | 1 / w]
__global__ __launch_bounds__(1024) void k_mu_final(const float* __restrict__ part, int64_t m, int d, int dpad8, float* __restrict__ mu8) {
    __shared__ float red[16];
    const int i = threadIdx.x;
    float mean = 0.f, var = 0.f;
    if (i < d) {
        float a = 0.f, a2 = 0.f;
        for (int p = 0; p < MU_PARTS; ++p) { a += part[((int64_t)p * 2) * dpad8 + i]; a2 += part[((int64_t)p * 2 + 1) * dpad8 + i]; }
        mean = a / (float)m;
        var = fmaxf(a2 / (float)m - mean * mean, 0.f);
        if (!(fabsf(mean) <= 3.4028235e38f) || !(var <= 3.4028235e38f)) { mean = 0.f; var = 0.f; }
    }
    float t = var;
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1) t += __shfl_xor(t, mm, 64);
    if ((i & 63) == 0) red[i >> 6] = t;
    __syncthreads();
    float tot = 0.f;
    for (int j = 0; j < 16; ++j) tot += red[j];
    const float rms = sqrtf(tot / (float)d);
    float w = 1.f;
    if (i < d && var > 0.f && rms > 0.f) {
        const float want = fminf(fmaxf(rms / sqrtf(var), 0.0625f), 16.f);
        w = exp2f(rintf(log2f(want)));
        if (!(w >= 0.0625f && w <= 16.f)) w = 1.f;
    }
    if (i < dpad8) { mu8[i] = mean; mu8[dpad8 + i] = w; mu8[2 * dpad8 + i] = 1.f / w; }
}

// rows [r0, r1) of the fp32 master -> int8 rows + scales, one wave per row: x - mu ~ sx8 x8; rows [r1, r_pad) become padding (scale NaN: never marked)
__global__ __launch_bounds__(256) void k_quant8_rows(const float* __restrict__ xf, const float* __restrict__ mu8, int8_t* __restrict__ x8, float* __restrict__ sx8,
                                                     int64_t r0, int64_t r1, int64_t r_pad, int d, int dpad8, float* __restrict__ bounds8) {
    const int lane = threadIdx.x & 63;
    const float* mu = mu8;
    const float* wt = mu8 + dpad8;
    float emax = 0.f, nmax = 0.f;
    bool bad = false;
    for (int64_t row = r0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < r_pad; row += (int64_t)gridDim.x * 4) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(x8 + row * dpad8);
        if (row >= r1) {
            for (int i = lane; i < dpad8 / 4; i += 64) dst[i] = 0u;
            if (lane == 0) sx8[row] = __uint_as_float(0x7fc00000u);
            continue;
        }
        const float* src = xf + row * d;
        auto centred = [&](int i) -> float4 {          // r = (x - mu) * w: one rounding (the subtraction); the product with a power of two is exact
            float4 v = *reinterpret_cast<const float4*>(src + i);
            const float4 c = *reinterpret_cast<const float4*>(mu + i), w = *reinterpret_cast<const float4*>(wt + i);
            v.x = (v.x - c.x) * w.x; v.y = (v.y - c.y) * w.y; v.z = (v.z - c.z) * w.z; v.w = (v.w - c.w) * w.w;
            return v;
        };
        float amax = 0.f, r2 = 0.f;
        bool fin = true;
        for (int i = lane * 4; i < d; i += 256) {
            const float4 v = centred(i);
            const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            fin = fin && (fabsf(v.x) <= 3.4028235e38f) && (fabsf(v.y) <= 3.4028235e38f) && (fabsf(v.z) <= 3.4028235e38f) && (fabsf(v.w) <= 3.4028235e38f);
            amax = fmaxf(amax, m);
            r2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, m, 64)); r2 += __shfl_xor(r2, m, 64); }
        fin = __all(fin);
        const float sx = fin ? amax / 127.f : 0.f;
        const float inv = (fin && amax > 0.f) ? 127.f / amax : 0.f;
        float e2 = 0.f;
        for (int i = lane * 4; i < dpad8; i += 256) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < d && fin) v = centred(i);
            const float a0 = fminf(fmaxf(rintf(v.x * inv), -127.f), 127.f), a1 = fminf(fmaxf(rintf(v.y * inv), -127.f), 127.f);
            const float a2 = fminf(fmaxf(rintf(v.z * inv), -127.f), 127.f), a3 = fminf(fmaxf(rintf(v.w * inv), -127.f), 127.f);
            const float d0 = v.x - sx * a0, d1 = v.y - sx * a1, d2 = v.z - sx * a2, d3 = v.w - sx * a3;
            e2 += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            dst[i >> 2] = ((uint32_t)(int)a0 & 255u) | (((uint32_t)(int)a1 & 255u) << 8) | (((uint32_t)(int)a2 & 255u) << 16) | (((uint32_t)(int)a3 & 255u) << 24);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) e2 += __shfl_xor(e2, m, 64);
        // slack: the fp32 roundings of x - mu, of sx * a and of the residual (each <= 2^-24 relative per component: together <= 3.6e-7 |r| in the norm,
        // |r| <= amax sqrt(d)) and of the sums themselves
        const float e = sqrtf(e2) * 1.0001f + 3.6e-7f * amax * sqrtf((float)d);
        const float rn = sqrtf(r2) * 1.0001f + 1.2e-7f * amax * sqrtf((float)d);
        if (fin && e == e && rn == rn) { emax = fmaxf(emax, e); nmax = fmaxf(nmax, rn); } else bad = true;
        if (lane == 0) sx8[row] = sx;
    }
    if (lane == 0) {
        atomicMax(reinterpret_cast<unsigned int*>(bounds8), __float_as_uint(emax));
        atomicMax(reinterpret_cast<unsigned int*>(bounds8) + 2, __float_as_uint(nmax));
        if (bad) atomicOr(reinterpret_cast<unsigned int*>(bounds8) + 1, 1u);
    }
}

// one block per query slot of the 32-query stream kernel: k-th best 16-bit score of the rows seen so far (the candidate buffer as k_select left it), the
// query's two byte planes, eps8, and the mark threshold in units of sx8 * (254 Ia + Ib).  Slots >= nq: zero planes, threshold +inf.
__global__ __launch_bounds__(256) void k_scan8_prep(const uint64_t* __restrict__ cand, int cand_cap, const uint32_t* __restrict__ cnt, const float* __restrict__ eps16,
                                                    const float* __restrict__ qf, int nq, int d, int dpad8, int k, const float* __restrict__ bounds8,
                                                    const float* __restrict__ mu8, int8_t* __restrict__ q8, float* __restrict__ thr8,
                                                    unsigned int* __restrict__ mark_count, float* __restrict__ thr16, uint32_t* __restrict__ cnt_spread, float eps8_scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ float red[5][4];
    if (blockIdx.x == 0 && threadIdx.x == 0) *mark_count = 0u;          // the list the scan behind this kernel appends to
    uint64_t* s = reinterpret_cast<uint64_t*>(smem);
    unsigned int* hist = reinterpret_cast<unsigned int*>(s + cand_cap);                 // 256 + 4
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int8_t* pa = q8 + (int64_t)q * dpad8;
    int8_t* pb = q8 + (int64_t)(32 + q) * dpad8;
    if (q >= nq) {
        for (int i = tid; i < dpad8; i += 256) { pa[i] = 0; pb[i] = 0; }
        if (tid == 0) thr8[q] = INFINITY;
        return;
    }
    int m = (int)cnt[q];
    if (tid == 0) cnt_spread[(int64_t)q * CNT_STRIDE] = (uint32_t)m;      // (before the clamp: an overflowed buffer stays visible as one)
    if (m > cand_cap) m = cand_cap;
    float kth = -INFINITY;
    if (m >= k) {
        const uint64_t* c = cand + (int64_t)q * cand_cap;
        for (int i = tid; i < m; i += 256) s[i] = c[i];
        __syncthreads();
        kth = ord_f32(radix_select_desc<256>(s, m, k, hist, tid));
    }
    const float* qv = qf + (int64_t)q * d;
    const float* mu = mu8;
    const float* winv = mu8 + 2 * dpad8;
    float amax = 0.f;
    for (int i = tid; i < d; i += 256) amax = fmaxf(amax, fabsf(qv[i] * winv[i]));      // u = q / w (exact: w is a power of two)
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1) amax = fmaxf(amax, __shfl_xor(amax, mm, 64));
    if (lane == 0) red[0][wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    const float sq = amax / 127.f;
    const float inv = amax > 0.f ? 127.f / amax : 0.f;
    float e2 = 0.f, u2 = 0.f, qm = 0.f, qma = 0.f;
    for (int i = tid; i < dpad8; i += 256) {
        const float v = i < d ? qv[i] : 0.f;
        const float u = v * winv[i];
        const float t = u * inv;
        const float a = fminf(fmaxf(rintf(t), -127.f), 127.f);
        const float b = fminf(fmaxf(rintf(254.f * (t - a)), -127.f), 127.f);
        const float uh = sq * (a + b * (1.f / 254.f));
        e2 += (u - uh) * (u - uh);
        u2 += u * u;
        const float c = mu[i];
        qm += v * c; qma += fabsf(v * c);
        pa[i] = (a == a) ? (int8_t)(int)a : (int8_t)0;
        pb[i] = (b == b) ? (int8_t)(int)b : (int8_t)0;
    }
#pragma unroll
    for (int mm = 32; mm >= 1; mm >>= 1) {
        e2 += __shfl_xor(e2, mm, 64); u2 += __shfl_xor(u2, mm, 64); qm += __shfl_xor(qm, mm, 64); qma += __shfl_xor(qma, mm, 64);
    }
    if (lane == 0) { red[1][wave] = e2; red[2][wave] = u2; red[3][wave] = qm; red[4][wave] = qma; }
    __syncthreads();
    if (tid == 0) {
        e2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        u2 = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        qm = red[3][0] + red[3][1] + red[3][2] + red[3][3];
        qma = red[4][0] + red[4][1] + red[4][2] + red[4][3];
        const float un = sqrtf(u2) * 1.0001f;
        const float du = sqrtf(e2) * 1.0001f + 4.8e-7f * un;               // |u - u^|, + the fp32 rounding of u^ itself in the residual above
        const float dx = bounds8[0];                                       // max |r - r^|,  r = (x - mu) w,  r^ = sx8 x8
        const float rn = bounds8[2] + dx;                                  // max |r^| <= max |r| + dx
        // q.x = q.mu + u.r;  |u.r - u^.r^| <= |u| dx + |u - u^| |r^|;  q.mu in fp32: any order of <= d additions, <= (d + 2) 2^-24 sum |q_i mu_i|;
        // the device's arithmetic on the byte score (one fma rounding, the scale product, the threshold's own scaling): <= 4 x 2^-23 relative
        float e8 = un * dx + du * rn + 4.8e-7f * un * rn + (float)(d + 2) * 6.0e-8f * qma * 1.01f;
        // + the three fp32 subtractions that form theta below, and a few ulps of distance from the canonical (fp32-rounded) scores of the k rows the bound
        // rests on: an unmarked row must not even TIE with them (a tie would be decided by the row index)
        e8 += 1e-6f * (fabsf(kth) + fabsf(qm) + eps16[q]);
        e8 = e8 * 1.001f + 1e-30f;
        e8 *= eps8_scale;                                                  // 1 in production; kr_set_option("debug_eps8_permille") lets a test run with a bound that is too small
        const float theta = kth - eps16[q] - e8 - qm;                      // bound on the byte score u^.r^ of a row that can still matter
        const bool usable = (fabsf(theta) <= 3.4028235e38f) && sq > 0.f && (sq <= 3.4028235e38f) && reinterpret_cast<const unsigned int*>(bounds8)[1] == 0u;
        float t = -INFINITY;                                               // unusable: every row is marked (slow, exact)
        if (usable) { t = theta * (254.f / sq); t -= fabsf(t) * 1e-6f; }
        thr8[q] = t;
        // the 16-bit threshold of the round as well: a row whose 16-bit score is below kth16 - 2 eps16 has an exact score below kth16 - eps16, i.e. below k rows
        // already seen — for small k that is far above the round's own survivor-budget threshold (the rank-55 score of the sample): a few hundred listed rows
        // per query pass instead of ~2000 (each costs a returning atomic on the query's counter, and k_rerank reads them all).  Kept strictly below what
        // k_rerank will compute from the same kth16 (theta = b_k - 2 eps with b_k >= kth16), so its "theta > thr" test still passes when nothing better turns up.
        float t16 = kth - 2.f * eps16[q];
        t16 -= fabsf(t16) * 1e-6f + 1e-30f;
        if (t16 > thr16[q]) thr16[q] = t16;                                // (NaN / -inf: no change)
    }
}

// The candidate counters of a block's queries are adjacent words: returning atomics on them execute one after the other in ONE memory channel (measured: ~6 ns each,
// 0.39 ms of k_score_list for 32 queries x ~2000 appended rows).  While the list is scored the counters therefore live CNT_STRIDE words apart (k_scan8_prep copies
// them out, k_cnt_fold copies them back): the queries' appends then proceed in parallel.
__global__ void k_cnt_fold(uint32_t* __restrict__ cnt, const uint32_t* __restrict__ spread, int nq) {
    const int q = threadIdx.x;
    if (q < nq) cnt[q] = spread[(int64_t)q * CNT_STRIDE];
}

// 16-bit scores of the marked rows, each for the queries that marked it (qmask[row], cleared on the way) -> candidate buffers: what the final round of the
// 16-bit scan does for the rows it streams.  One wave per listed row and step, the next row and its mask requested before the current one is scored; the
// queries' 16-bit copies sit in LDS.  The sum order differs from the MFMA chain's; eps16 covers any order of at most dpad fp32 additions (k_prep_queries:
// 1.1 dpad 2^-24 |qc| |xc|; here 16 sequential + 6 tree steps).  dpad <= 1024.
template <class T>
__global__ __launch_bounds__(1024) void k_score_list(const uint16_t* __restrict__ xc, int dpad, const uint16_t* __restrict__ qc, int nq,
                                                     const uint32_t* __restrict__ rowlist, const unsigned int* __restrict__ count, uint32_t* __restrict__ qmask,
                                                     const float* __restrict__ thr, uint32_t* __restrict__ cnt, int cnt_stride, uint64_t* __restrict__ cand, int cand_cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* qs = reinterpret_cast<uint16_t*>(smem);                     // [nq][dpad]
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned wpb = blockDim.x >> 6;                                 // waves per block
    const unsigned n = *count;
    const unsigned W = gridDim.x * wpb;
    unsigned i = blockIdx.x * wpb + (unsigned)(tid >> 6);
    if (blockIdx.x * wpb >= n) return;                                    // (block-uniform) nothing for this block: skip the query load too
    for (int j = tid * 8; j < nq * dpad; j += (int)blockDim.x * 8) *reinterpret_cast<uint4*>(qs + j) = *reinterpret_cast<const uint4*>(qc + j);
    __syncthreads();
    // a lane holds 8 consecutive elements of each 512-element half of the row
    uint4 cur[2], nxt[2];
    uint32_t row = 0u, nrow = 0u, m = 0u, nm = 0u;
    auto load = [&](uint4 (&r)[2], uint32_t rw) {
        const uint16_t* xr = xc + (int64_t)rw * dpad + lane * 8;
        r[0] = (lane * 8 < dpad) ? *reinterpret_cast<const uint4*>(xr) : make_uint4(0u, 0u, 0u, 0u);
        r[1] = (lane * 8 + 512 < dpad) ? *reinterpret_cast<const uint4*>(xr + 512) : make_uint4(0u, 0u, 0u, 0u);
    };
    auto dot8 = [&](const uint4& v, const uint16_t* qq, float acc) -> float {
        const uint4 u = *reinterpret_cast<const uint4*>(qq);
        acc = fmaf(T::to_f32((uint16_t)(v.x & 0xffffu)), T::to_f32((uint16_t)(u.x & 0xffffu)), acc); acc = fmaf(T::to_f32((uint16_t)(v.x >> 16)), T::to_f32((uint16_t)(u.x >> 16)), acc);
        acc = fmaf(T::to_f32((uint16_t)(v.y & 0xffffu)), T::to_f32((uint16_t)(u.y & 0xffffu)), acc); acc = fmaf(T::to_f32((uint16_t)(v.y >> 16)), T::to_f32((uint16_t)(u.y >> 16)), acc);
        acc = fmaf(T::to_f32((uint16_t)(v.z & 0xffffu)), T::to_f32((uint16_t)(u.z & 0xffffu)), acc); acc = fmaf(T::to_f32((uint16_t)(v.z >> 16)), T::to_f32((uint16_t)(u.z >> 16)), acc);
        acc = fmaf(T::to_f32((uint16_t)(v.w & 0xffffu)), T::to_f32((uint16_t)(u.w & 0xffffu)), acc); acc = fmaf(T::to_f32((uint16_t)(v.w >> 16)), T::to_f32((uint16_t)(u.w >> 16)), acc);
        return acc;
    };
    if (i < n) { row = rowlist[i]; m = qmask[row]; load(cur, row); }
    for (; i < n; i += W) {
        const bool more = i + W < n;
        if (more) { nrow = rowlist[i + W]; nm = qmask[nrow]; load(nxt, nrow); }
        if (lane == 0) qmask[row] = 0u;                                  // this row's mask has been read (by this wave only: a row is listed once)
        uint32_t mm = __builtin_amdgcn_readfirstlane(m);
        while (mm) {
            const int q = __ffs(mm) - 1;
            mm &= mm - 1u;
            float acc = 0.f;
            if (lane * 8 < dpad) acc = dot8(cur[0], qs + q * dpad + lane * 8, acc);
            if (lane * 8 + 512 < dpad) acc = dot8(cur[1], qs + q * dpad + 512 + lane * 8, acc);
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) acc += __shfl_xor(acc, sft, 64);
            if (lane == 0 && acc >= thr[q]) {
                const unsigned pos = atomicAdd(&cnt[(int64_t)q * cnt_stride], 1u);
                if (pos < (unsigned)cand_cap) cand[(int64_t)q * cand_cap + pos] = make_key(acc, row);
            }
        }
        cur[0] = nxt[0]; cur[1] = nxt[1]; row = nrow; m = nm;
    }
}

// the status record of one query block, written straight into the call's pinned host record (one kernel instead of three to five 4-byte .. 4-KiB copies,
// each of which is a 9-us copy kernel of its own on the stream): [QBLK] flags | [QBLK] re-ranked rows | list overflow | rows marked by the byte pre-scan
// (BYTE_UNUSED: not taken) | non-finite-row flag of the int8 copy
constexpr uint32_t BYTE_UNUSED = 0xffffffffu;
__global__ __launch_bounds__(256) void k_status_pack(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ nrer, const unsigned int* __restrict__ list_overflow,
                                                     const unsigned int* __restrict__ marked, const float* __restrict__ bounds8, int nq, uint32_t* __restrict__ rec) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nq) { rec[i] = flags[i]; rec[QBLK + i] = nrer[i]; }
    if (i == 0) {
        rec[2 * QBLK] = *list_overflow;
        rec[2 * QBLK + 1] = marked ? *marked : BYTE_UNUSED;
        rec[2 * QBLK + 2] = marked ? reinterpret_cast<const uint32_t*>(bounds8)[1] : 0u;
    }
}

// ---------------------------------------------------------------------------------------------------------
// exact full scan (fallback for flagged queries, and mode = 1)
// ---------------------------------------------------------------------------------------------------------
// grid (nchunks, nqf): canonical score of every row of the chunk for one query, chunk-local top-kk keys
template <int NCH>   // NCH 256-element steps cover a row: d <= 256 NCH
__global__ __launch_bounds__(256) void k_exact_scan(const float* __restrict__ qf, const int* __restrict__ qidx, const float* __restrict__ xf,
                                                    int64_t n, int d, int kk, uint64_t* __restrict__ out, int nchunks, int rc, int force_exact) {
    // rc = rows per block (a power of two, kk <= rc <= EXACT_RC): 1024 for corpus-sized scans, as little as 64 for a transient candidate set of a
    // few hundred rows (kr_score_topk), so that the rows of ONE query are spread over several CUs instead of being a 128-step latency chain on one
    __shared__ uint64_t s[EXACT_RC];
    __shared__ unsigned long long exact_limbs[4][EXACT_NLIMB];
    const int chunk = blockIdx.x, qi = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* qv = qf + (int64_t)qidx[qi] * d;
    const int64_t r0 = (int64_t)chunk * rc;
    float4 qr[NCH];                                            // the query stays in registers; two rows per wave step, all their loads issued first
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int i = lane * 4 + j * 256;
        qr[j] = (i < d) ? *reinterpret_cast<const float4*>(qv + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // as in k_rerank: the next pair of rows is requested before the current pair is scored
    RowPair<NCH> cur, nxt;
    if (r0 + wave < n) load_rows2<NCH>(cur, xf + (r0 + wave) * d, xf + ((r0 + wave + 4 < n && wave + 4 < rc) ? r0 + wave + 4 : r0 + wave) * d, d, lane);
    for (int i = wave; i < rc; i += 8) {
        const int64_t row0 = r0 + i, row1 = r0 + i + 4;
        const bool v0 = row0 < n, v1 = row1 < n && i + 4 < rc;
        const int in = i + 8;
        const int64_t nrow0 = r0 + in, nrow1 = r0 + in + 4;
        const bool nv0 = in < rc && nrow0 < n, nv1 = nrow1 < n && in + 4 < rc;
        if (nv0) load_rows2<NCH>(nxt, xf + nrow0 * d, xf + (nv1 ? nrow1 : nrow0) * d, d, lane);
        float e0 = 0.f, e1 = 0.f;
        if (v0) score_rows2<NCH>(qr, qv, cur, xf + row0 * d, xf + (v1 ? row1 : row0) * d, d, lane, exact_limbs[wave], force_exact != 0, e0, e1);
        cur = nxt;
        if (lane == 0) {
            uint64_t k0 = v0 ? make_key(e0, (uint32_t)row0) : 0ull, k1 = v1 ? make_key(e1, (uint32_t)row1) : 0ull;
            if (v0 && k0 == 0ull) k0 = 1ull;   // cannot happen for row < 2^32-1; keeps "0 = padding" unambiguous
            if (v1 && k1 == 0ull) k1 = 1ull;
            s[i] = k0;
            if (i + 4 < rc) s[i + 4] = k1;
        }
    }
    bitonic_sort_desc(s, rc, tid, 256);
    uint64_t* o = out + ((int64_t)qi * nchunks + chunk) * kk;
    for (int j = tid; j < kk; j += 256) o[j] = s[j];
}

// grid (nchunks_out, nqf): sort SORT_CHUNK keys of in[qi][...], write the kk best
__global__ __launch_bounds__(256) void k_sort_chunks(const uint64_t* __restrict__ in, int64_t m_in, int kk, uint64_t* __restrict__ out, int nchunks_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem);
    const int chunk = blockIdx.x, qi = blockIdx.y, tid = threadIdx.x;
    const int64_t base = (int64_t)chunk * SORT_CHUNK;
    for (int i = tid; i < SORT_CHUNK; i += 256) s[i] = (base + i < m_in) ? in[(int64_t)qi * m_in + base + i] : 0ull;
    bitonic_sort_desc(s, SORT_CHUNK, tid, 256);
    uint64_t* o = out + ((int64_t)qi * nchunks_out + chunk) * kk;
    for (int j = tid; j < kk; j += 256) o[j] = s[j];
}

__global__ void k_keys_to_out(const uint64_t* __restrict__ keys, int64_t stride, const int* __restrict__ qidx, int nqf, int k,
                              float* __restrict__ out_s, int64_t* __restrict__ out_r) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nqf * k) return;
    const int qi = i / k, j = i % k;
    const uint64_t key = keys[(int64_t)qi * stride + j];
    const int q = qidx[qi];
    out_s[(int64_t)q * k + j] = key_score(key);
    out_r[(int64_t)q * k + j] = (int64_t)key_row(key);
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
// rows added by an asynchronous kr_index_add (device source: only ev_add was recorded) must have landed before anything reads or moves them on
// ANOTHER stream: the blocking copies below run on the NULL stream, which does not order against a caller's non-blocking stream
static int wait_adds_host(Index* ix) {
    if (ix->ev_add) KR_HIP(hipEventSynchronize(ix->ev_add));
    return 0;
}

static unsigned long long va_retired_total() { return g_va_retired.load() + g_va_retired_bias.load(); }
constexpr unsigned long long VA_BUDGET = 48ull << 40;         // of the 47-bit address space; beyond it new indexes grow by hipMalloc + copy

// Move the index into a fresh pair of mapped ranges that can hold `want` rows (reserve: 8 x want, at least 16 GiB of rows, at most what the device
// could ever hold), copying the n rows present.  Nothing of *ix changes unless every step succeeded.  -1: mechanism unavailable, KR_E*: hard error.
static int vmm_move(Index* ix, int64_t want) {
    const size_t row_f = (size_t)ix->d * sizeof(float), row_c = (size_t)ix->dpad * 2;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || total_b == 0) { (void)hipGetLastError(); return -1; }
    const int64_t max_rows = (int64_t)(total_b / (row_f + row_c)) + 4096;          // more rows than the device could ever hold
    int64_t res_rows = std::max<int64_t>(8 * want, (int64_t)((size_t)g_vmm_min_reserve.load() / (row_f + row_c)));
    res_rows = std::min(std::max(res_rows, want), std::max(max_rows, want));
    VBuf nf, nc;
    if (nf.init(ix->device, (size_t)res_rows * row_f) != 0 || nc.init(ix->device, (size_t)res_rows * row_c) != 0) { nf.release(); nc.release(); return -1; }
    const int64_t ncap = round_up(want, 256);
    const int rf = nf.ensure((size_t)ncap * row_f), rc = rf == 0 ? nc.ensure((size_t)ncap * row_c) : rf;
    if (rf != 0 || rc != 0) {
        nf.release(); nc.release();
        if (rf == KR_ENOMEM || rc == KR_ENOMEM) return fail(KR_ENOMEM, "out of device memory growing the index to %lld rows", (long long)want);
        return -1;
    }
    if (ix->n > 0) {   // includes the NaN padding rows of the 16-bit copy
        hipError_t e = hipMemcpy(nf.base, ix->xf, (size_t)ix->n * row_f, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(nc.base, ix->xc, (size_t)round_up(ix->n, 256) * row_c, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { nf.release(); nc.release(); return fail(KR_EHIP, "moving the index rows failed: %s", hipGetErrorString(e)); }
    }
    if (ix->vmm == 1) { (void)hipDeviceSynchronize(); ix->vf.release(); ix->vc.release(); }
    else { if (ix->xf) (void)hipFree(ix->xf); if (ix->xc) (void)hipFree(ix->xc); }
    ix->vf = nf; ix->vc = nc; ix->vmm = 1;
    ix->xf = reinterpret_cast<float*>(ix->vf.base); ix->xc = reinterpret_cast<uint16_t*>(ix->vc.base);
    ix->cap_rows = (int64_t)std::min(ix->vf.mapped / row_f, ix->vc.mapped / row_c) / 256 * 256;
    return 0;
}

static void drop_byte_copy(Index* ix);
static int grow_once(Index* ix, int64_t want);
//  the rows come first: when the device is out of memory and the index holds its derived int8 copy, that copy is released and the growth is
// tried once more — an add that succeeded before the copy existed must not fail because of it
static int grow(Index* ix, int64_t want) {
    const int rc = grow_once(ix, want);
    if (rc != KR_ENOMEM || !ix->x8) return rc;
    (void)hipGetLastError();
    drop_byte_copy(ix);
    return grow_once(ix, want);
}
static int grow_once(Index* ix, int64_t want) {
    if (want <= ix->cap_rows) return 0;
    const size_t row_f = (size_t)ix->d * sizeof(float), row_c = (size_t)ix->dpad * 2;
    KR_TRY(wait_adds_host(ix));
    // small indexes (< 256 MiB of rows) live in plain hipMalloc memory and grow by copy; from 256 MiB on the rows move ONCE into mapped chunks
    // (64 MiB each) inside a reserved address range of max(16 GiB, 8 x the requested rows), where they stay until that range is used up (then: ONE
    // more move into a range 8 x larger, i.e. the whole device — a 100-GB index appended without reserve() copies its first 16 GiB once).  Round 2
    // reserved the whole HBM for every index: ~170 create / destroy cycles used up the address budget below; now > 3000 for indexes up to 16 GiB.
    if (ix->vmm < 0 && (size_t)want * (row_f + row_c) >= ((size_t)256 << 20)) {
        ix->vmm = 0;
        if (ix->no_vmm) {
        } else if (va_retired_total() >= VA_BUDGET) {
            static std::atomic<bool> said{false};
            if (!said.exchange(true)) fprintf(stderr, "kirag_amd: %.1f TiB of virtual addresses retired by earlier indexes; new indexes grow by hipMalloc + copy\n",
                                              (double)va_retired_total() / (double)(1ull << 40));
        } else {
            const int rc = vmm_move(ix, want);
            if (rc == 0) return 0;
            if (rc != -1) return rc;
        }
    } else if (ix->vmm == 1) {
        const int64_t ncap = round_up(want, 256);
        if ((size_t)ncap * row_f <= ix->vf.reserved && (size_t)ncap * row_c <= ix->vc.reserved) {
            const int rf = ix->vf.ensure((size_t)ncap * row_f), rc = rf == 0 ? ix->vc.ensure((size_t)ncap * row_c) : rf;
            if (rf == 0 && rc == 0) {
                ix->cap_rows = (int64_t)std::min(ix->vf.mapped / row_f, ix->vc.mapped / row_c) / 256 * 256;
                return 0;                                                       // rows did not move; the NaN padding behind row n is still in place
            }
            if (rf == KR_ENOMEM || rc == KR_ENOMEM) return fail(KR_ENOMEM, "out of device memory growing the index to %lld rows", (long long)want);
            return fail(KR_EHIP, "mapping more index memory failed");
        }
        const int rc = vmm_move(ix, want);                                      // reservation used up: one move into a larger one
        if (rc == 0) return 0;
        return rc != -1 ? rc : fail(KR_EHIP, "reserving a larger address range for the index failed");
    }
    int64_t ncap = std::max<int64_t>(want, ix->cap_rows + ix->cap_rows / 2);
    ncap = round_up(ncap, 256);
    float* nf = nullptr; uint16_t* nc = nullptr;
    if (hipMalloc(&nf, (size_t)ncap * ix->d * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return fail(KR_ENOMEM, "hipMalloc of the index rows failed (%lld rows)", (long long)ncap); }
    hipError_t e = hipMalloc(&nc, (size_t)ncap * ix->dpad * 2);
    if (e != hipSuccess) { (void)hipFree(nf); return fail(KR_ENOMEM, "hipMalloc of the coarse copy failed: %s", hipGetErrorString(e)); }
    if (ix->n > 0) {
        e = hipMemcpy(nf, ix->xf, (size_t)ix->n * ix->d * sizeof(float), hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(nc, ix->xc, (size_t)ix->n * ix->dpad * 2, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { (void)hipFree(nf); (void)hipFree(nc); return fail(KR_EHIP, "copying the index rows failed: %s", hipGetErrorString(e)); }
    }
    if (ix->xf) (void)hipFree(ix->xf);
    if (ix->xc) (void)hipFree(ix->xc);
    ix->xf = nf; ix->xc = nc; ix->cap_rows = ncap;
    const int64_t n_pad = round_up(ix->n, 256);
    if (n_pad > ix->n) {   // keep the NaN padding rows behind the last row (see k_coarse)
        hipLaunchKernelGGL(k_pad_nan, dim3((unsigned)(((n_pad - ix->n) * ix->dpad + 255) / 256)), dim3(256), 0, 0, ix->xc, ix->n, n_pad, ix->dpad);
        KR_HIP(hipDeviceSynchronize());
    }
    return 0;
}

static int ensure_ws(Index* ix, int k, int cand_cap) {
    if (!ix->q_f) {
        KR_HIP(hipMalloc(&ix->q_f, (size_t)QBLK * ix->d * sizeof(float)));
        KR_HIP(hipMalloc(&ix->q_c, (size_t)QBLK * ix->dpad * 2));
        KR_HIP(hipMalloc(&ix->thr, QBLK * sizeof(float)));
        KR_HIP(hipMalloc(&ix->eps, QBLK * sizeof(float)));
        KR_HIP(hipMalloc(&ix->cnt, QBLK * sizeof(uint32_t)));
        KR_HIP(hipMalloc(&ix->flags, QBLK * sizeof(uint32_t)));
        KR_HIP(hipMalloc(&ix->nrer, QBLK * sizeof(uint32_t)));
        KR_HIP(hipMalloc(&ix->q_f2, (size_t)32 * ix->d * sizeof(float)));
        KR_HIP(hipMalloc(&ix->theta1, (size_t)THETA_BLOCKS * QBLK * sizeof(float)));
        KR_HIP(hipMalloc(&ix->thr_mark, 32 * sizeof(float)));
        KR_HIP(hipMalloc(&ix->ex_qidx, QBLK * sizeof(int)));
        KR_HIP(hipMalloc(&ix->blk_list, (size_t)ix->num_cu * ShapeC::NWAVE * WLISTCAP * sizeof(uint4)));
        KR_HIP(hipMalloc(&ix->blk_cnt, ((size_t)ix->num_cu * ShapeC::NWAVE + 4) * sizeof(unsigned int)));
        for (auto& e : ix->ev) KR_HIP(hipEventCreate(&e));
        for (auto& e : ix->evc) KR_HIP(hipEventCreate(&e));
    }
    if (cand_cap > ix->cand_cap) {
        if (ix->cand) (void)hipFree(ix->cand);
        ix->cand = nullptr; ix->cand_cap = 0;
        KR_HIP(hipMalloc(&ix->cand, (size_t)QBLK * cand_cap * sizeof(uint64_t)));
        ix->cand_cap = cand_cap;
    }
    if (k > ix->out_k) {
        if (ix->out_s) (void)hipFree(ix->out_s);
        if (ix->out_r) (void)hipFree(ix->out_r);
        ix->out_s = nullptr; ix->out_r = nullptr; ix->out_k = 0;
        KR_HIP(hipMalloc(&ix->out_s, (size_t)QBLK * k * sizeof(float)));
        KR_HIP(hipMalloc(&ix->out_r, (size_t)QBLK * k * sizeof(int64_t)));
        ix->out_k = k;
    }
    return 0;
}

static int64_t gcd64(int64_t a, int64_t b) { while (b) { int64_t t = a % b; a = b; b = t; } return a; }

// exact scan of the queries listed in ix->ex_qidx[0..nqf) -> ix->out_s / out_r rows of those queries
static int exact_scan(Index* ix, int nqf, int k, hipStream_t st) {
    // rows per block: 1024, or fewer for small row counts so that at least ~16 blocks share the rows of a query (never below k or 64)
    int rc = EXACT_RC;
    while (rc > 64 && rc / 2 >= k && ix->n < (int64_t)rc * 16) rc /= 2;
    const int64_t nchunks = (ix->n + rc - 1) / rc;
    const int kk = std::min<int>(k, rc);
    // queries are processed in groups so that the ping/pong buffers stay bounded (<= 256 MiB each)
    const int64_t per_q = nchunks * kk;
    int group = (int)std::max<int64_t>(1, std::min<int64_t>(nqf, (int64_t)(32u << 20) / std::max<int64_t>(per_q, 1)));
    const size_t need = (size_t)group * per_q * sizeof(uint64_t);
    if (need > ix->ex_bytes) {
        if (ix->ex_a) (void)hipFree(ix->ex_a);
        if (ix->ex_b) (void)hipFree(ix->ex_b);
        ix->ex_a = ix->ex_b = nullptr; ix->ex_bytes = 0;
        KR_HIP(hipMalloc(&ix->ex_a, need));
        KR_HIP(hipMalloc(&ix->ex_b, need));
        ix->ex_bytes = need;
    }
    for (int g0 = 0; g0 < nqf; g0 += group) {
        const int g = std::min(group, nqf - g0);
        auto scan = ix->d <= 1024 ? &k_exact_scan<4> : ix->d <= 2048 ? &k_exact_scan<8> : &k_exact_scan<16>;
        hipLaunchKernelGGL(scan, dim3((unsigned)nchunks, g), dim3(256), 0, st, ix->q_f, ix->ex_qidx + g0, ix->xf, ix->n, ix->d, kk,
                           ix->ex_a, (int)nchunks, rc, ix->force_exact);
        uint64_t* cur = ix->ex_a; uint64_t* nxt = ix->ex_b;
        int64_t m = per_q;
        while (m > kk) {   // reduce until one sorted list of kk keys per query remains
            const int64_t nco = (m + SORT_CHUNK - 1) / SORT_CHUNK;
            hipLaunchKernelGGL(k_sort_chunks, dim3((unsigned)nco, g), dim3(256), SORT_CHUNK * sizeof(uint64_t), st, cur, m, kk, nxt, (int)nco);
            std::swap(cur, nxt);
            m = nco * kk;
        }
        if (per_q <= kk) {   // a single chunk: already sorted by k_exact_scan
        }
        hipLaunchKernelGGL(k_keys_to_out, dim3((g * k + 255) / 256), dim3(256), 0, st, cur, m, ix->ex_qidx + g0, g, k, ix->out_s, ix->out_r);
    }
    KR_HIP(hipGetLastError());
    return 0;
}

// once per (process, device): hipFuncSetAttribute for the kernels that need more than 64 KiB of dynamic LDS.  Function attributes belong to
// the device's code-object instance; std::call_once makes the guard safe when several host threads search on different handles.
struct DevOnce { std::once_flag f[64]; int rc[64] = {}; };
template <class F> static int once_per_device(DevOnce& o, int device, F&& fn) {
    const int i = device & 63;
    std::call_once(o.f[i], [&] { o.rc[i] = fn(); });
    return o.rc[i];
}

template <class T, int KT>
static int launch_q32_kt(const CoarseArgs& a, int num_cu, int device, hipStream_t st) {
    constexpr int lds = q32_lds<KT>();
    static DevOnce once;
    KR_TRY(once_per_device(once, device, [&]() -> int {
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coarse_q32<T, 1, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coarse_q32<T, 0, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coarse_q32<T, 2, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        return 0;
    }));
    if (a.direct == 2) hipLaunchKernelGGL((k_coarse_q32<T, 2, KT>), dim3(num_cu), dim3(Q32_THREADS), lds, st, a);
    else if (a.direct) hipLaunchKernelGGL((k_coarse_q32<T, 1, KT>), dim3(num_cu), dim3(Q32_THREADS), lds, st, a);
    else hipLaunchKernelGGL((k_coarse_q32<T, 0, KT>), dim3(num_cu), dim3(Q32_THREADS), lds, st, a);
    return 0;
}
template <class T>
static int launch_q32(const CoarseArgs& a, int kt64, int num_cu, int device, hipStream_t st) {
    switch (kt64) {
        case 16: return launch_q32_kt<T, 16>(a, num_cu, device, st);
        case 12: return launch_q32_kt<T, 12>(a, num_cu, device, st);
        case 8: return launch_q32_kt<T, 8>(a, num_cu, device, st);
        case 6: return launch_q32_kt<T, 6>(a, num_cu, device, st);
    }
    return fail(KR_EINVAL, "no k_coarse_q32 instance for dpad/64 = %d", kt64);
}

// ---- byte pre-scan: host side ---------------------------------------------------------------------------------------------------------------------------
constexpr int BYTE_NQ_MAX = 32;                   // the whole range of the <= 32-query stream kernel (k_score_list keeps the queries' 16-bit copies in <= 64 KiB of LDS)

template <int KT>
static int launch_scan8_kt(const CoarseArgs& a, int num_cu, int device, hipStream_t st) {
    constexpr int lds = q32_lds<KT>();
    static DevOnce once;
    KR_TRY(once_per_device(once, device, [&]() -> int {
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coarse_q32<BF16, 3, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        return 0;
    }));
    hipLaunchKernelGGL((k_coarse_q32<BF16, 3, KT>), dim3(num_cu), dim3(Q32_THREADS), lds, st, a);
    return 0;
}
static bool byte_dim_ok(int d) { const int dp = (int)round_up(d, 128); return dp == 1024 || dp == 768 || dp == 512 || dp == 384; }

// the bitmap / row list shared by pass 2's marking scan and the byte pre-scan (one bit per row + one word: the list length; the marked rows, compacted)
static int ensure_bitmap(Index* ix) {
    size_t words = (size_t)((ix->n + 31) / 32);
    if (words <= ix->bitmap_words) return 0;
    // sized for the rows the index has room for (at least 1.5 x the previous size): kr_index_prepare runs after every append of a streamed build
    // (Indexer.index_data), and a re-allocation - hipFree waits for the device - per 512-row batch would stall that pipeline
    words = std::max(words, std::max((size_t)((ix->cap_rows + 31) / 32), ix->bitmap_words + ix->bitmap_words / 2));
    if (ix->bitmap) (void)hipFree(ix->bitmap);
    if (ix->rowlist) (void)hipFree(ix->rowlist);
    if (ix->qmask) (void)hipFree(ix->qmask);
    ix->bitmap = nullptr; ix->rowlist = nullptr; ix->qmask = nullptr; ix->bitmap_words = 0;
    KR_HIP(hipMalloc(&ix->bitmap, (words + 1) * sizeof(uint32_t)));
    KR_HIP(hipMalloc(&ix->rowlist, words * 32 * sizeof(uint32_t)));
    KR_HIP(hipMalloc(&ix->qmask, words * 32 * sizeof(uint32_t)));
    KR_HIP(hipMemset(ix->bitmap, 0, (words + 1) * sizeof(uint32_t)));        // from here on k_compact_rows leaves it all-zero behind every scan
    KR_HIP(hipMemset(ix->qmask, 0, words * 32 * sizeof(uint32_t)));          // ... and k_score_list the per-row query masks
    ix->bitmap_words = words;
    ix->cnt_word_at = (size_t)-1;
    return 0;
}

// the list-length word sits right behind the bits of the CURRENT row count; when rows were added since its last use the old word (a stale count) now lies inside
// the bitmap proper, which every scan expects to be all-zero: clear it (stream-ordered, 4 bytes, only when n changed)
static int move_count_word(Index* ix, size_t words, hipStream_t st) {
    if (ix->cnt_word_at != words) {
        if (ix->cnt_word_at != (size_t)-1 && ix->cnt_word_at <= ix->bitmap_words) KR_HIP(hipMemsetAsync(ix->bitmap + ix->cnt_word_at, 0, sizeof(uint32_t), st));
        ix->cnt_word_at = words;
    }
    return 0;
}

// The int8 copy is DERIVED data (1 KiB per row at d = 1024): whenever the index stops using it (non-finite rows, an allocation failure) or needs the memory for
// the rows themselves (grow), it is released; a later small search may build it again unless byte_off is set.  hipFree waits for the device: searches that read
// the copy are done.
static void drop_byte_copy(Index* ix) {
    if (ix->x8) (void)hipFree(ix->x8);
    if (ix->sx8) (void)hipFree(ix->sx8);
    ix->x8 = nullptr; ix->sx8 = nullptr; ix->n8 = 0; ix->cap8 = 0;
}

// HBM budget of the copy (it used to be allocated without looking): building it must leave max(2 GiB, 1/16 of the device) free for
// the rows still to come, the search workspaces and the caller's own tensors; otherwise the index simply keeps answering small blocks from the 16-bit copy.
static bool byte_copy_fits(size_t bytes) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return true; }
    const size_t keep = std::max<size_t>((size_t)2 << 30, total_b / 16);
    return free_b >= bytes + keep;
}

// Bring the int8 copy up to row n (kernels on `st`, behind the adds the caller has already ordered).  true: usable for this call.  An allocation failure is
// not an error: the copy is an accelerator, the index then serves every block from the 16-bit copy (byte_off).
static bool ensure_byte_copy(Index* ix, hipStream_t st) {
    if (ix->byte_off) return false;
    const int dpad8 = (int)round_up(ix->d, 128);
    auto give_up = [&]() { (void)hipGetLastError(); ix->byte_off = true; drop_byte_copy(ix); return false; };
    if (!ix->thr8) {
        ix->dpad8 = dpad8;
        if (!ix->bounds8 && hipMalloc(&ix->bounds8, 4 * sizeof(float)) != hipSuccess) return give_up();
        if (!ix->q8 && hipMalloc(&ix->q8, (size_t)2 * 32 * dpad8) != hipSuccess) return give_up();
        if (hipMemsetAsync(ix->bounds8, 0, 4 * sizeof(float), st) != hipSuccess) return give_up();
        if (!ix->mu8 && hipMalloc(&ix->mu8, (size_t)(3 + 2 * MU_PARTS) * dpad8 * sizeof(float)) != hipSuccess) return give_up();
        if (!ix->cnt_spread && hipMalloc(&ix->cnt_spread, (size_t)32 * CNT_STRIDE * sizeof(uint32_t)) != hipSuccess) return give_up();
        if (hipMalloc(&ix->thr8, 32 * sizeof(float)) != hipSuccess) return give_up();
    }
    if (ix->n8 == ix->n) return true;
    const int64_t need = round_up(ix->n, 256);
    if (need > ix->cap8) {
        if (ix->x8) (void)hipFree(ix->x8);            // (hipFree waits for the device: earlier searches that read the old copy are done)
        if (ix->sx8) (void)hipFree(ix->sx8);
        ix->x8 = nullptr; ix->sx8 = nullptr; ix->n8 = 0;
        const int64_t ncap = std::max(need, round_up(ix->cap8 + ix->cap8 / 2, 256));
        ix->cap8 = 0;
        if (!byte_copy_fits((size_t)ncap * (dpad8 + sizeof(float)))) return false;      // not now (byte_off stays clear: memory may be free again later)
        if (hipMalloc(&ix->x8, (size_t)ncap * dpad8) != hipSuccess) return give_up();
        if (hipMalloc(&ix->sx8, (size_t)ncap * sizeof(float)) != hipSuccess) return give_up();
        ix->cap8 = ncap;
        if (hipMemsetAsync(ix->bounds8, 0, 4 * sizeof(float), st) != hipSuccess) return give_up();
    }
    if (ix->n8 == 0) {   // a fresh copy: its centre = the mean of the first rows (fixed from here on: later rows are quantised around the same centre)
        const int64_t m = std::min<int64_t>(ix->n, MU_ROWS);
        hipLaunchKernelGGL(k_mu_partial, dim3(MU_PARTS), dim3(256), 0, st, ix->xf, m, ix->d, dpad8, ix->mu8 + 3 * dpad8);
        hipLaunchKernelGGL(k_mu_final, dim3(1), dim3(1024), 0, st, ix->mu8 + 3 * dpad8, m, ix->d, dpad8, ix->mu8);
    }
    const int64_t rows = need - ix->n8;
    const unsigned grid = (unsigned)std::min<int64_t>((rows + 3) / 4, (int64_t)ix->num_cu * 16);
    hipLaunchKernelGGL(k_quant8_rows, dim3(grid), dim3(256), 0, st, ix->xf, ix->mu8, ix->x8, ix->sx8, ix->n8, ix->n, need, ix->d, dpad8, ix->bounds8);
    if (hipGetLastError() != hipSuccess) return give_up();
    ix->n8 = ix->n;
    return true;
}

// The final round of a small block through the int8 copy: thresholds + query planes, the marking stream over the round's tile slots, the row list (the
// compaction also clears the bits it reads), the 16-bit scores of the listed rows into the candidate buffers.  Enqueue only; the list length stays on the device.
template <class T>
static int byte_final_round(Index* ix, const CoarseArgs& a, const float* qf, int nq, int k, hipStream_t st) {
    const size_t words = (size_t)((ix->n + 31) / 32);
    KR_TRY(move_count_word(ix, words, st));
    unsigned int* cnt_word = ix->bitmap + words;                         // (the list-length word behind the bitmap pass 2's marking scan uses)
    const size_t prep_lds = (size_t)ix->cand_cap * sizeof(uint64_t) + 264 * sizeof(unsigned int);
    hipLaunchKernelGGL(k_scan8_prep, dim3(32), dim3(256), prep_lds, st, ix->cand, ix->cand_cap, ix->cnt, ix->eps, qf, nq, ix->d, ix->dpad8, k, ix->bounds8, ix->mu8,
                       ix->q8, ix->thr8, cnt_word, ix->thr, ix->cnt_spread, (float)g_eps8_permille.load() * 1e-3f);
    CoarseArgs m = a;
    m.xc = reinterpret_cast<const uint16_t*>(ix->x8); m.dpad = ix->dpad8 / 2;
    m.qc = reinterpret_cast<const uint16_t*>(ix->q8); m.qc2 = reinterpret_cast<const uint16_t*>(ix->q8 + (size_t)32 * ix->dpad8);
    m.sx8 = ix->sx8; m.thr = ix->thr8; m.bitmap = ix->bitmap; m.qmask = ix->qmask; m.nq_pad = 32; m.nq = nq; m.direct = 3;
    switch (ix->dpad8 / 128) {
        case 8: KR_TRY(launch_scan8_kt<8>(m, ix->num_cu, ix->device, st)); break;
        case 6: KR_TRY(launch_scan8_kt<6>(m, ix->num_cu, ix->device, st)); break;
        case 4: KR_TRY(launch_scan8_kt<4>(m, ix->num_cu, ix->device, st)); break;
        case 3: KR_TRY(launch_scan8_kt<3>(m, ix->num_cu, ix->device, st)); break;
        default: return fail(KR_EINVAL, "no byte pre-scan instance for d = %d", ix->d);
    }
    hipLaunchKernelGGL(k_compact_rows<true>, dim3((unsigned)((words + 1023) / 1024)), dim3(1024), 0, st, ix->bitmap, (int64_t)words, ix->rowlist, cnt_word);
    // few queries: many small blocks (each reloads <= 16 KiB of queries); up to 32: 1024-thread blocks, two per CU, around 64 KiB of queries each
    const size_t sl_lds = (size_t)nq * ix->dpad * sizeof(uint16_t);
    static DevOnce sl_once;
    KR_TRY(once_per_device(sl_once, ix->device, [&]() -> int {
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score_list<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 1024 * 2));
        return 0;
    }));
    uint32_t* const cnt = nq > 1 ? ix->cnt_spread : ix->cnt;            // (one query: nothing to spread)
    const int cstride = nq > 1 ? CNT_STRIDE : 1;
    if (nq <= 8) hipLaunchKernelGGL(k_score_list<T>, dim3(ix->num_cu * 8), dim3(256), sl_lds, st, ix->xc, ix->dpad, ix->q_c, nq, ix->rowlist, cnt_word, ix->qmask,
                                    ix->thr, cnt, cstride, ix->cand, ix->cand_cap);
    else hipLaunchKernelGGL(k_score_list<T>, dim3(ix->num_cu * 2), dim3(1024), sl_lds, st, ix->xc, ix->dpad, ix->q_c, nq, ix->rowlist, cnt_word, ix->qmask,
                            ix->thr, cnt, cstride, ix->cand, ix->cand_cap);
    if (nq > 1) hipLaunchKernelGGL(k_cnt_fold, dim3(1), dim3(64), 0, st, ix->cnt, ix->cnt_spread, nq);
    KR_HIP(hipGetLastError());
    return 0;
}

// over-fetch K1 and buffer capacity for top-k: K1 = pow2 >= 2.5 k while that is <= 512 (k <= 204), then pow2 >= 1.25 k (k <= 1638 > the k limit);
// cap = min(16 K1, 8192) candidates per query (k_select / k_rerank keep the buffer in LDS).  The round growth factor follows from cap / K1.
static void plan_buffers(int k, int& K1, int& cap, int& rmax) {
    K1 = std::max(64, next_pow2((5 * k + 1) / 2));
    if (K1 > 512) K1 = std::max(1024, next_pow2((5 * k + 3) / 4));
    cap = std::min(16 * K1, 8192);
    rmax = cap >= 8192 ? 8192 : RERANK_MAX;          // candidates the certified re-rank can take (LDS: cap + rmax keys)
}

// One filter / select pipeline over queries [0, nq) of the workspace arrays (thr, cnt, flags, cand): `launch(a)` runs one scan round over the
// tile slots [a.tile_begin, a.tile_begin + a.tile_count) (slot = bm rows), k_select tightens the thresholds between rounds.
// Round schedule (any thresholds are SAFE: the certificate in k_rerank decides exactness; the schedule only sets speed):
//   round 0: cap rows, every score stored (direct slots);  growth rounds: g x the rows seen (g = cap / (2 K1) - 1: expected survivors cap / 2),
//   thr = K1-th best; as soon as rows_seen * (cap / 64) >= rows_left the rest is ONE final round whose threshold is the r-th best seen with
//   r = (cap/2) * seen / left  (expected cap/2 survivors), 32 <= r <= K1.
// `final_round` (may be null): called INSTEAD of launch() for the last round when that round is not the direct one and covers at least half of the rows
// (the byte pre-scan of small blocks, byte_final_round); *final_used tells the caller whether it was.
template <class Launch>
static int run_rounds(Index* ix, CoarseArgs& a, int64_t n_rows, int nq, int bm, int K1, int cap, hipStream_t st, hipEvent_t* timed, Launch&& launch, int& final_preset, int& rounds,
                      const std::function<int(const CoarseArgs&)>* final_round = nullptr, bool* final_used = nullptr) {
    a.ntiles = (n_rows + bm - 1) / bm;
    // interleaving permutation: multiplier near ntiles / golden ratio, coprime to ntiles
    int64_t mul = std::max<int64_t>(1, (int64_t)((double)a.ntiles * 0.6180339887498949));
    while (gcd64(mul, a.ntiles) != 1) ++mul;
    a.perm_mul = mul % a.ntiles; if (a.perm_mul == 0) a.perm_mul = 1;
    const size_t sel_lds = (size_t)ix->cand_cap * sizeof(uint64_t) + 264 * sizeof(unsigned int);
    const int64_t growth = std::max(1, cap / (2 * K1) - 1);
    int64_t done = 0;
    int64_t step = std::max<int64_t>(1, cap / bm);
    int round = 0;
    final_preset = 0;
    a.direct = 1;
    while (done < a.ntiles) {
        const int64_t cnt_t = std::min<int64_t>(step, a.ntiles - done);
        a.tile_begin = done; a.tile_count = cnt_t;
        if (timed && round < 16) KR_HIP(hipEventRecord(timed[2 * round], st));
        if (final_round && !a.direct && done + cnt_t >= a.ntiles && cnt_t * bm * 2 >= n_rows) {
            KR_TRY((*final_round)(a));
            if (final_used) *final_used = true;
        } else KR_TRY(launch(a));
        if (timed && round < 16) KR_HIP(hipEventRecord(timed[2 * round + 1], st));
        const int preset = a.direct ? (int)(cnt_t * bm) : 0;
        ++round;
        done += cnt_t;
        a.direct = 0;
        if (done >= a.ntiles) { final_preset = preset; break; }   // last round: k_rerank reads the buffer as it is; thr stays
        const int64_t seen = done * bm, left = n_rows - seen;
        int rank = K1;
        // final round only once the rank it needs is >= 32 WITHOUT clamping: (cap/2) * seen / left >= 32.  (With a fixed "seen * 64 >= left" the
        // small buffer of k <= 25 (cap = 1024) got rank 32 with up to 32 * 64 = 2048 expected survivors: overflow -> fallback.)
        if (seen * std::max(1, cap / 64) >= left) {
            step = a.ntiles - done;
            rank = (int)std::min<int64_t>(K1, std::max<int64_t>(32, ((int64_t)(cap / 2) * seen + left - 1) / left));
            rank = std::min(rank, K1);
        } else {
            step = done * growth;
        }
        if (nq <= 32) hipLaunchKernelGGL(k_select<1024>, dim3(nq), dim3(1024), sel_lds, st, ix->cand, ix->cand_cap, ix->cnt, ix->flags, ix->thr, K1, rank, preset);
        else hipLaunchKernelGGL(k_select<256>, dim3(nq), dim3(256), sel_lds, st, ix->cand, ix->cand_cap, ix->cnt, ix->flags, ix->thr, K1, rank, preset);
    }
    rounds = round;
    return 0;
}

// device-usable aliases of a small call's own query / result buffers (device memory of this GPU or pinned host memory): the kernels then read the queries
// and write the results in place — no staging copy through the workspace (three ~10-us copy kernels per one-query search).  All null: the staged path.
struct DirectIO { const float* q = nullptr; float* scores = nullptr; int64_t* rows = nullptr; };
static const void* device_alias(const void* p, int device) {
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (a.type == hipMemoryTypeDevice) return a.device == device ? p : nullptr;
    if (a.type == hipMemoryTypeHost) return a.devicePointer;          // pinned: mapped into the device's address space
    return nullptr;
}

static int launch_rerank(Index* ix, int nq, int k, int rmax, int final_preset, const float* qf, const int* qmap, float* theta_out, hipStream_t st,
                         const float* theta_ext = nullptr, float* out_s = nullptr, int64_t* out_r = nullptr) {
    const size_t rer_lds = (size_t)ix->cand_cap * sizeof(uint64_t) + (size_t)rmax * sizeof(uint64_t) + 264 * sizeof(unsigned int);
    auto rerank = ix->d <= 1024 ? &k_rerank<4> : ix->d <= 2048 ? &k_rerank<8> : &k_rerank<16>;   // row steps held in registers
    hipLaunchKernelGGL(rerank, dim3(nq), dim3(256), rer_lds, st, ix->cand, ix->cand_cap, ix->cnt, ix->flags, ix->thr, ix->eps, qf, ix->xf, ix->d, k,
                       final_preset, rmax, out_s ? out_s : ix->out_s, out_r ? out_r : ix->out_r, ix->nrer, qmap, ix->force_exact, theta_out, theta_ext);
    KR_HIP(hipGetLastError());
    return 0;
}

constexpr int P2_WORDS = 3 * QBLK + 16;      // pass-2 status region (h_status): the slow passes run synchronously, one call at a time
static int ensure_status(Index* ix, int) {
    if (ix->status_blocks) return 0;
    KR_HIP(hipHostMalloc(reinterpret_cast<void**>(&ix->h_status), (size_t)P2_WORDS * sizeof(uint32_t), hipHostMallocDefault));
    ix->status_blocks = 1;
    return 0;
}
// the per-block status records of one outstanding call
static int ensure_slot_status(Index::Pending& pd, int nblocks) {
    if (!pd.ev_done) KR_HIP(hipEventCreateWithFlags(&pd.ev_done, hipEventDisableTiming));
    if (nblocks <= pd.status_blocks) return 0;
    const int nb = std::max(nblocks, 4);
    if (pd.status) (void)hipHostFree(pd.status);
    pd.status = nullptr; pd.status_blocks = 0;
    KR_HIP(hipHostMalloc(reinterpret_cast<void**>(&pd.status), (size_t)nb * STATUS_STRIDE * sizeof(uint32_t), hipHostMallocDefault));
    pd.status_blocks = nb;
    return 0;
}

struct BlockPlan { bool smallq, q32; int kt64, nq_pad, K1, cap, rmax; };
static BlockPlan plan_block(const Index* ix, int nq, int k, bool byte_block = false) {
    BlockPlan p;
    p.smallq = nq <= ShapeSplit::BN;             // one 128-query tile: HBM-bound scan on the producer / consumer loop
    p.kt64 = ix->dpad / 64;
    p.q32 = nq <= 32 && (p.kt64 == 16 || p.kt64 == 12 || p.kt64 == 8 || p.kt64 == 6) && !ix->no_q32;   // register-resident queries, pure stream
    p.nq_pad = p.q32 ? 32 : (int)round_up(nq, p.smallq ? ShapeSplit::BN : ShapeC::BN);   // <= QBLK = 1024 = 4 x 256
    plan_buffers(k, p.K1, p.cap, p.rmax);
    // a block whose final round streams the int8 copy wants that round to start early (every 16-bit round before it costs a launch plus 2 bytes per
    // element): with 4096 candidates per query the schedule is the direct round (4096 rows), one 16-bit round of 31 x that, then the byte round over
    // the remaining ~97 % (with 1024 candidates the final round cannot start before 6 % of the rows have been seen: run_rounds' survivor budget)
    if (byte_block && p.cap < 4096) p.cap = 4096;
    return p;
}

static int search_attrs(Index* ix) {
    static DevOnce sel_once;
    return once_per_device(sel_once, ix->device, [&]() -> int {   // cap = 8192 needs 64 KiB + of dynamic LDS
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select<256>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8 + 264 * 4));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_select<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8 + 264 * 4));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_local_topk), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8 + 264 * 4));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_scan8_prep), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8 + 264 * 4));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_global_theta), hipFuncAttributeMaxDynamicSharedMemorySize, MERGE_MAX * 8 + 264 * 4));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rerank<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8 + 8192 * 8 + 264 * 4));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rerank<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8 + 8192 * 8 + 264 * 4));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rerank<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 8 + 8192 * 8 + 264 * 4));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fine<true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fine<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fine<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fine<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return 0;
    });
}

static void fill_args(const Index* ix, CoarseArgs& a) {
    a = CoarseArgs{};
    a.xc = ix->xc; a.n = ix->n; a.dpad = ix->dpad; a.qc = ix->q_c;
    a.thr = ix->thr; a.cnt = ix->cnt; a.flags = ix->flags; a.cand = ix->cand; a.cand_cap = ix->cand_cap;
    a.blk_list = ix->blk_list; a.blk_cnt = ix->blk_cnt; a.list_overflow = ix->blk_cnt + ix->num_cu * ShapeC::NWAVE;
    a.xf = ix->xf; a.d = ix->d; a.bitmap = nullptr; a.rowlist = nullptr; a.nlist = 0;
}

// ---- pass 1 of one block of <= 1024 queries: 16-bit MFMA scan + certified re-rank.  ENQUEUE ONLY: the per-query certificate flags, the re-rank counts and
// the list-overflow word go to the block's status record in pinned memory, the (optimistic) results straight into the caller's buffers; nothing here waits
// for the device.  kr_index_search_finish reads the status records and re-answers the flagged queries (slow_passes).
static int pass1_rerank(Index* ix, int nq, int k, int rmax, int final_preset, float* scores, int64_t* rows, int blk, hipStream_t st, uint32_t* status,
                        const float* theta_ext, bool byte_used, const DirectIO& dio = DirectIO{});
template <class T>
static int pass1_coarse(Index* ix, const float* q, int nq, int k, int blk, hipStream_t st, int& rounds, int& final_preset, int& rmax, bool& byte_used,
                        const float* q_direct = nullptr);
template <class T>
static int pass1_enqueue(Index* ix, const float* q, int nq, int k, float* scores, int64_t* rows, int blk, hipStream_t st, int& rounds, uint32_t* status,
                         const DirectIO& dio = DirectIO{}) {
    int final_preset = 0, rmax = 0;
    bool byte_used = false;
    KR_TRY(pass1_coarse<T>(ix, q, nq, k, blk, st, rounds, final_preset, rmax, byte_used, dio.q));
    return pass1_rerank(ix, nq, k, rmax, final_preset, scores, rows, blk, st, status, nullptr, byte_used, dio);
}

// first half: queries -> 16-bit copy + bounds, the coarse rounds; leaves the candidate buffers of the block in the workspace
template <class T>
static int pass1_coarse(Index* ix, const float* q, int nq, int k, int blk, hipStream_t st, int& rounds, int& final_preset, int& rmax, bool& byte_used,
                        const float* q_direct) {
    // small block on a large index: the final round goes through the int8 copy (byte_final_round)
    bool byte_ok = plan_block(ix, nq, k).q32 && nq <= BYTE_NQ_MAX && !ix->byte_off && g_byte_prescan.load() != 0 && ix->n >= (int64_t)g_byte_min_rows.load() && byte_dim_ok(ix->d);
    if (byte_ok && ix->byte_pause > 0) { --ix->byte_pause; byte_ok = false; }
    if (byte_ok) byte_ok = ensure_byte_copy(ix, st);
    if (byte_ok && ensure_bitmap(ix) != 0) {        // 8 B per row of bitmap / row list / query masks: without them the 16-bit round serves the block
        (void)hipGetLastError();
        byte_ok = false; ix->byte_off = true; drop_byte_copy(ix);
    }
    const BlockPlan p = plan_block(ix, nq, k, byte_ok);
    rmax = p.rmax;
    byte_used = false;
    KR_TRY(ensure_ws(ix, k, p.cap));
    KR_TRY(search_attrs(ix));
    if (!q_direct) KR_HIP(hipMemcpyAsync(ix->q_f, q, (size_t)nq * ix->d * sizeof(float), hipMemcpyDefault, st));
    const float* const qsrc = q_direct ? q_direct : ix->q_f;
    CoarseArgs a; fill_args(ix, a);
    const size_t blk_cnt_bytes = ((size_t)ix->num_cu * ShapeC::NWAVE + 4) * sizeof(unsigned int);
    hipLaunchKernelGGL(k_prep_queries<T>, dim3(p.nq_pad), dim3(64), 0, st, qsrc, ix->q_c, nq, ix->d, ix->dpad, ix->bounds, ix->eps, ix->thr,
                       ix->cnt, ix->flags, ix->blk_cnt, (int)(blk_cnt_bytes / sizeof(unsigned int)));
    a.nq_pad = p.nq_pad; a.nq = nq;
    const int bm = p.q32 ? 32 : p.smallq ? ShapeSplit::BM : ShapeC::BM;
    const int lds = p.smallq ? COARSE_LDS_SMALLQ : COARSE_LDS;
    static DevOnce coarse_once;
    KR_TRY(once_per_device(coarse_once, ix->device, [&]() -> int {
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coarse<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, COARSE_LDS));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coarse<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, COARSE_LDS));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coarse<T, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, COARSE_LDS_SMALLQ));
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_coarse<T, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, COARSE_LDS_SMALLQ));
        return 0;
    }));
    final_preset = 0;
    rounds = 0;
    const std::function<int(const CoarseArgs&)> byte_round = [&](const CoarseArgs& ca) -> int { return byte_final_round<T>(ix, ca, qsrc, nq, k, st); };
    KR_TRY(run_rounds(ix, a, ix->n, nq, bm, p.K1, p.cap, st, blk < TIMED_BLOCKS ? ix->evc + blk * 32 : nullptr, [&](const CoarseArgs& ca) -> int {
        if (p.q32) return launch_q32<T>(ca, p.kt64, ix->num_cu, ix->device, st);
        if (p.smallq) {
            if (ca.direct) hipLaunchKernelGGL((k_coarse<T, true, true>), dim3(ix->num_cu), dim3(ShapeC::NTHREADS), lds, st, ca);
            else hipLaunchKernelGGL((k_coarse<T, false, true>), dim3(ix->num_cu), dim3(ShapeC::NTHREADS), lds, st, ca);
        } else if (ca.direct) hipLaunchKernelGGL((k_coarse<T, true>), dim3(ix->num_cu), dim3(ShapeC::NTHREADS), lds, st, ca);
        else hipLaunchKernelGGL((k_coarse<T, false>), dim3(ix->num_cu), dim3(ShapeC::NTHREADS), lds, st, ca);
        return 0;
    }, final_preset, rounds, byte_ok ? &byte_round : nullptr, &byte_used));
    ix->st.coarse_rounds += rounds;
    return 0;
}

// second half: exactness certificate + exact re-rank of the candidates, status records and results on their way to the caller
static int pass1_rerank(Index* ix, int nq, int k, int rmax, int final_preset, float* scores, int64_t* rows, int blk, hipStream_t st, uint32_t* status,
                        const float* theta_ext, bool byte_used, const DirectIO& dio) {
    KR_TRY(launch_rerank(ix, nq, k, rmax, final_preset, dio.q ? dio.q : ix->q_f, nullptr, blk < THETA_BLOCKS ? ix->theta1 + (size_t)blk * QBLK : nullptr, st, theta_ext,
                         dio.scores, dio.rows));
    uint32_t* rec = status + (size_t)blk * STATUS_STRIDE;
    hipLaunchKernelGGL(k_status_pack, dim3((nq + 255) / 256), dim3(256), 0, st, ix->flags, ix->nrer, ix->blk_cnt + ix->num_cu * ShapeC::NWAVE,
                       byte_used ? ix->bitmap + (size_t)((ix->n + 31) / 32) : nullptr, ix->bounds8, nq, rec);
    KR_HIP(hipGetLastError());
    if (!dio.scores) {
        KR_HIP(hipMemcpyAsync(scores, ix->out_s, (size_t)nq * k * sizeof(float), hipMemcpyDefault, st));
        KR_HIP(hipMemcpyAsync(rows, ix->out_r, (size_t)nq * k * sizeof(int64_t), hipMemcpyDefault, st));
    }
    return 0;
}

// ---- passes 2 and 3 for the queries of one block whose hflags entry is non-zero (blocking: the rare path).  `restore`: the workspace no longer holds this
// block (a later block of the same call went through it): its queries are copied in again and, when pass 1 produced results, its result rows too.
template <class T>
static int slow_passes(Index* ix, const float* q, int nq, int k, float* scores, int64_t* rows, int blk, hipStream_t st,
                       const std::vector<uint32_t>& hflags, bool coarse_pass, bool fine_allowed, bool restore) {
    const BlockPlan p = plan_block(ix, nq, k);
    const int K1 = p.K1, cap = p.cap, rmax = p.rmax, kt64 = p.kt64;
    const bool fine_ok = fine_allowed && ix->d <= FINE_DMAX && !ix->no_fine;
    KR_TRY(ensure_ws(ix, k, cap));
    KR_TRY(search_attrs(ix));
    if (restore) {
        KR_HIP(hipMemcpyAsync(ix->q_f, q, (size_t)nq * ix->d * sizeof(float), hipMemcpyDefault, st));
        if (coarse_pass) {
            KR_HIP(hipMemcpyAsync(ix->out_s, scores, (size_t)nq * k * sizeof(float), hipMemcpyDefault, st));
            KR_HIP(hipMemcpyAsync(ix->out_r, rows, (size_t)nq * k * sizeof(int64_t), hipMemcpyDefault, st));
        }
    }
    CoarseArgs a; fill_args(ix, a);
    const size_t blk_cnt_bytes = ((size_t)ix->num_cu * ShapeC::NWAVE + 4) * sizeof(unsigned int);
    std::vector<int> fl;
    for (int i = 0; i < nq; ++i) {
        if (hflags[i]) { fl.push_back(i); if (coarse_pass && (hflags[i] & 1u)) ix->st.overflow++; }
    }
    const int64_t n_flagged = (int64_t)fl.size();
    if (fl.empty()) return 0;
    uint32_t* const hs = ix->h_status;                      // pass-2 status region
    // ---- pass 2: fp64 MFMA scan of the fp32 master rows for the flagged queries, in groups that share one pass over the corpus -------------------
    std::vector<int> fl3;
    if (fine_ok) {
        const int gmax = ix->d <= 1024 ? FINE_QMAX : 16;
        KR_HIP(hipMemcpyAsync(ix->ex_qidx, fl.data(), fl.size() * sizeof(int), hipMemcpyHostToDevice, st));
        KR_HIP(hipEventRecord(ix->ev[1], st));
        int ngroups = 0;
        bool mark_useful = true;
        for (size_t g0 = 0; g0 < fl.size(); g0 += gmax, ++ngroups) {
            const int g = (int)std::min<size_t>(gmax, fl.size() - g0);
            const int nt = g <= 16 ? 1 : 2;
            const int* qmap = ix->ex_qidx + g0;
            hipLaunchKernelGGL(k_gather_rows, dim3(g), dim3(256), 0, st, ix->q_f, qmap, ix->q_f2, ix->d);
            // pre-scan: one HBM-bound pass of the 16-bit copy with the group's queries in registers (the <= 32-query stream kernel, MODE 2) sets one bit
            // per ROW that reaches theta1 = b_k - 2 eps of pass 1 for some query of the group; a row below that bound is strictly below the query's
            // k-th exact score (the k best coarse rows all have exact >= b_k - eps), so the fp64 scan only has to visit the marked rows, which
            // k_compact_rows turns into a list (k_fine addresses its rows through it).  A boilerplate cluster of 2 % of the corpus, scattered over the
            // whole row range, thus costs one 1.6-ms stream + 2 % of the fp64 work instead of the full 6-ms fp64 pass; a corpus in which every row
            // is marked costs the stream on top.
            a.rowlist = nullptr; a.nlist = 0;
            int64_t n_scan = ix->n;
            // (once a group's pre-scan marks more than a quarter of the rows, the remaining groups of this call skip it: on such data it only adds its stream)
            const bool can_mark = coarse_pass && blk < THETA_BLOCKS && mark_useful && (kt64 == 16 || kt64 == 12 || kt64 == 8 || kt64 == 6) && !ix->no_mark;
            if (can_mark) {
                const size_t words = (size_t)((ix->n + 31) / 32);
                KR_TRY(ensure_bitmap(ix));                                              // (+ 1 word: the list length lives behind the bits)
                KR_TRY(move_count_word(ix, words, st));
                KR_HIP(hipMemsetAsync(ix->bitmap, 0, (words + 1) * sizeof(uint32_t), st));
                hipLaunchKernelGGL(k_prep_queries<T>, dim3(32), dim3(64), 0, st, ix->q_f2, ix->q_c, g, ix->d, ix->dpad, ix->bounds, ix->eps, ix->thr, ix->cnt, ix->flags);
                hipLaunchKernelGGL(k_gather_theta, dim3(1), dim3(64), 0, st, ix->theta1 + (size_t)blk * QBLK, qmap, g, ix->thr_mark);
                CoarseArgs m = a;
                m.qc = ix->q_c; m.nq_pad = 32; m.nq = g; m.thr = ix->thr_mark; m.bitmap = ix->bitmap; m.direct = 2;
                m.ntiles = (ix->n + 31) / 32; m.perm_mul = 1; m.tile_begin = 0; m.tile_count = m.ntiles;
                KR_TRY((launch_q32<T>(m, kt64, ix->num_cu, ix->device, st)));
                unsigned int* cnt_word = ix->bitmap + words;
                hipLaunchKernelGGL(k_compact_rows<true>, dim3((unsigned)((words + 1023) / 1024)), dim3(1024), 0, st, ix->bitmap, (int64_t)words, ix->rowlist, cnt_word);
                KR_HIP(hipMemcpyAsync(hs + 3 * QBLK + 8, cnt_word, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
                KR_HIP(hipStreamSynchronize(st));                 // the row count sizes the rounds of this group (pass 2 is the rare path)
                const int64_t marked = (int64_t)hs[3 * QBLK + 8];
                if (marked >= k && marked <= ix->n / 2) { a.rowlist = ix->rowlist; a.nlist = marked; n_scan = marked; }
                if (marked > ix->n / 4) mark_useful = false;
                ix->st.marked_passes++; ix->st.marked_rows += marked;
            }
            hipLaunchKernelGGL(k_prep_fine<0>, dim3(16 * nt), dim3(64), 0, st, ix->q_f2, g, ix->d, ix->bounds, ix->eps, ix->thr, ix->cnt, ix->flags);
            KR_HIP(hipMemsetAsync(ix->blk_cnt, 0, blk_cnt_bytes, st));
            a.nq_pad = 16 * nt; a.nq = g; a.qf = ix->q_f2;
            const int lds = fine_lds_bytes(ix->d, 16 * nt);
            int final_preset = 0, rounds = 0;
            KR_TRY(run_rounds(ix, a, n_scan, g, 32, K1, cap, st, nullptr, [&](const CoarseArgs& ca) -> int {
                if (nt == 1) {
                    if (ca.direct) hipLaunchKernelGGL((k_fine<true, 1>), dim3(ix->num_cu), dim3(FINE_THREADS), lds, st, ca);
                    else hipLaunchKernelGGL((k_fine<false, 1>), dim3(ix->num_cu), dim3(FINE_THREADS), lds, st, ca);
                } else {
                    if (ca.direct) hipLaunchKernelGGL((k_fine<true, 2>), dim3(ix->num_cu), dim3(FINE_THREADS), lds, st, ca);
                    else hipLaunchKernelGGL((k_fine<false, 2>), dim3(ix->num_cu), dim3(FINE_THREADS), lds, st, ca);
                }
                return 0;
            }, final_preset, rounds));
            ix->st.fine_rounds += rounds;
            KR_TRY(launch_rerank(ix, g, k, rmax, final_preset, ix->q_f2, qmap, nullptr, st));
            KR_HIP(hipMemcpyAsync(hs + g0, ix->flags, g * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            KR_HIP(hipMemcpyAsync(hs + QBLK + g0, ix->nrer, g * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            KR_HIP(hipMemcpyAsync(hs + 2 * QBLK + ngroups, ix->blk_cnt + ix->num_cu * ShapeC::NWAVE, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
        }
        KR_HIP(hipEventRecord(ix->ev[2], st));
        KR_HIP(hipStreamSynchronize(st));
        for (size_t j = 0; j < fl.size(); ++j) {
            const bool ovf = hs[2 * QBLK + j / gmax] != 0u;
            ix->st.reranked_rows += hs[QBLK + j];
            if (hs[j] || ovf) fl3.push_back(fl[j]);
        }
        float fms = 0.f;
        if (hipEventElapsedTime(&fms, ix->ev[1], ix->ev[2]) == hipSuccess) ix->st.last_fine_ms += fms;
        ix->st.fine += (int64_t)fl.size() - (int64_t)fl3.size();
    } else {
        fl3 = fl;
    }
    // ---- pass 3: exact scan, one query at a time (mass ties beyond the re-rank capacity, NaN shortages, d > FINE_DMAX, mode 1) -----------------------
    if (!fl3.empty()) {
        KR_HIP(hipMemcpyAsync(ix->ex_qidx, fl3.data(), fl3.size() * sizeof(int), hipMemcpyHostToDevice, st));
        KR_TRY(exact_scan(ix, (int)fl3.size(), k, st));
        ix->st.exact += (int64_t)fl3.size();
    }
    KR_HIP(hipMemcpyAsync(scores, ix->out_s, (size_t)nq * k * sizeof(float), hipMemcpyDefault, st));
    KR_HIP(hipMemcpyAsync(rows, ix->out_r, (size_t)nq * k * sizeof(int64_t), hipMemcpyDefault, st));
    KR_HIP(hipStreamSynchronize(st));
    ix->st.fallback += n_flagged;
    return 0;
}

// enqueue pass 1 of every block of a search into the next free slot of the pending ring; the slot describes the call until it has been finished
static int finish_one(Index* ix, int64_t* flagged_out);
static int begin_search(Index* ix, const float* q, int nq, int k, float* scores, int64_t* rows, hipStream_t st) {
    const int nblocks = (nq + QBLK - 1) / QBLK;
    KR_TRY(ensure_status(ix, nblocks));
    // the device workspace is shared by all calls in STREAM order: outstanding calls of another stream are finished first, and so is the oldest one
    // when the ring is full
    while (ix->pend_n > 0 && (ix->pend[ix->pend_head].st != st || ix->pend_n == Index::PEND_MAX)) KR_TRY(finish_one(ix, nullptr));
    if (ix->ev_add) KR_HIP(hipStreamWaitEvent(st, ix->ev_add, 0));   // rows added asynchronously on another stream
    ix->force_exact = g_force_exact.load();
    ix->st.last_coarse_ms = 0.0; ix->st.last_total_ms = 0.0; ix->st.last_fine_ms = 0.0;
    Index::Pending& pd = ix->pend[(ix->pend_head + ix->pend_n) % Index::PEND_MAX];
    KR_TRY(ensure_slot_status(pd, nblocks));
    pd.q = q; pd.nq = nq; pd.k = k; pd.scores = scores; pd.rows = rows; pd.st = st; pd.rounds.assign(nblocks, 0);
    pd.seq = ++ix->call_seq;
    ix->ws_owner = pd.seq;
    {   // ensure_ws creates the events on first use: make sure they exist before the first record
        int K1, cap, rmax; plan_buffers(k, K1, cap, rmax);
        KR_TRY(ensure_ws(ix, k, cap));
    }
    KR_HIP(hipEventRecord(ix->ev[0], st));
    pd.active = true;      // from here on finish_one() has something to wait for, also when a later block fails to enqueue
    ++ix->pend_n;
    // a small call (the latency-bound case) whose buffers the device can address works in place
    DirectIO dio;
    pd.direct_io = false;
    if (nq <= 32) {
        dio.q = static_cast<const float*>(device_alias(q, ix->device));
        dio.scores = static_cast<float*>(const_cast<void*>(device_alias(scores, ix->device)));
        dio.rows = static_cast<int64_t*>(const_cast<void*>(device_alias(rows, ix->device)));
        if (!dio.q || !dio.scores || !dio.rows) dio = DirectIO{};
        pd.direct_io = dio.q != nullptr;
    }
    for (int b = 0; b < nblocks; ++b) {
        const int nb = std::min(QBLK, nq - b * QBLK);
        const float* qb = q + (size_t)b * QBLK * ix->d;
        float* sb = scores + (size_t)b * QBLK * k; int64_t* rb = rows + (size_t)b * QBLK * k;
        int rc;
        if (ix->coarse == KR_COARSE_BF16) rc = pass1_enqueue<BF16>(ix, qb, nb, k, sb, rb, b, st, pd.rounds[b], pd.status, dio);
        else rc = pass1_enqueue<F16>(ix, qb, nb, k, sb, rb, b, st, pd.rounds[b], pd.status, dio);
        if (rc) { (void)hipStreamSynchronize(st); pd.active = false; --ix->pend_n; return rc; }
    }
    KR_HIP(hipEventRecord(ix->ev[3], st));
    KR_HIP(hipEventRecord(pd.ev_done, st));
    return 0;
}

// the OLDEST outstanding call: wait for its pass 1, read its status records, run passes 2 / 3 for flagged queries (patching the caller's buffers),
// update the statistics.  *flagged_out = queries of the call that pass 1 could not certify (their rows in the caller's buffers were re-written).
static int finish_one(Index* ix, int64_t* flagged_out) {
    if (flagged_out) *flagged_out = 0;
    if (ix->pend_n == 0) return 0;
    Index::Pending& pd = ix->pend[ix->pend_head];
    ix->pend_head = (ix->pend_head + 1) % Index::PEND_MAX; --ix->pend_n;
    if (!pd.active) return 0;
    pd.active = false;
    if (pd.half) {      // the first half of a split search was enqueued and its second half never came: nothing to read, nothing the caller could use
        pd.half = false;
        (void)hipStreamSynchronize(pd.st);
        return fail(KR_ESTATE, "kr_index_search_coarse_async without its kr_index_search_rerank_async: the search was dropped");
    }
    KR_HIP(hipEventSynchronize(pd.ev_done));
    const bool newest = pd.seq == ix->call_seq;          // theta1 (pass 2's pre-scan bound) and the timing events hold the newest call's values
    const int nblocks = (pd.nq + QBLK - 1) / QBLK;
    int64_t flagged_total = 0;
    for (int b = 0; b < nblocks; ++b) {
        const int nb = std::min(QBLK, pd.nq - b * QBLK);
        const uint32_t* rec = pd.status + (size_t)b * STATUS_STRIDE;
        const bool list_ovf = rec[2 * QBLK] != 0u;   // a block list overflowed: every query of the block goes on to the next pass
        if (rec[2 * QBLK + 1] != BYTE_UNUSED) {
            // feedback of the byte pre-scan: an index with non-finite rows never takes it again; four blocks in a row that marked more than 1/8 of the
            // rows (data on which the byte bound does not separate) pause it for the next 1024 small blocks
            const int64_t marked = (int64_t)rec[2 * QBLK + 1];
            ix->st.byte_scans++; ix->st.byte_marked_rows += marked;
            if (rec[2 * QBLK + 2] != 0u) { ix->byte_off = true; if (ix->pend_n <= 1) drop_byte_copy(ix); }   // (a newer call in flight may still read it: kr_index_destroy frees it then)
            else if (marked > ix->n / 8) { if (++ix->byte_bad >= 4) { ix->byte_bad = 0; ix->byte_pause = 1024; } }
            else ix->byte_bad = 0;
        }
        std::vector<uint32_t> hflags(nb);
        int64_t nfl = 0;
        for (int i = 0; i < nb; ++i) { hflags[i] = rec[i] | (list_ovf ? 1u : 0u); ix->st.reranked_rows += rec[QBLK + i]; nfl += hflags[i] != 0u; }
        if (b < TIMED_BLOCKS && newest)
            for (int r = 0; r < pd.rounds[b] && r < 16; ++r) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ix->evc[b * 32 + 2 * r], ix->evc[b * 32 + 2 * r + 1]) == hipSuccess) ix->st.last_coarse_ms += ms;
            }
        if (nfl) {
            const float* qb = pd.q + (size_t)b * QBLK * ix->d;
            float* sb = pd.scores + (size_t)b * QBLK * pd.k; int64_t* rb = pd.rows + (size_t)b * QBLK * pd.k;
            const int tb = newest ? b : THETA_BLOCKS;     // an older call's theta1 has been overwritten: pass 2 without the pre-scan
            // ... and so have its queries and pass-1 results in the workspace (by a later call's pass 1, or by the slow passes of an older call that was
            // finished just before this one): copied in again
            const bool restore = nblocks > 1 || ix->ws_owner != pd.seq || pd.direct_io;
            ix->ws_owner = pd.seq;
            int rc;
            if (ix->coarse == KR_COARSE_BF16) rc = slow_passes<BF16>(ix, qb, nb, pd.k, sb, rb, tb, pd.st, hflags, true, true, restore);
            else rc = slow_passes<F16>(ix, qb, nb, pd.k, sb, rb, tb, pd.st, hflags, true, true, restore);
            if (rc) return rc;
        }
        flagged_total += nfl;
    }
    float tot = 0.f;
    if (newest && hipEventElapsedTime(&tot, ix->ev[0], ix->ev[3]) == hipSuccess) ix->st.last_total_ms += tot;
    ix->st.queries += pd.nq;
    ix->st.certified += pd.nq - flagged_total;
    if (flagged_out) *flagged_out = flagged_total;
    return 0;
}

// every outstanding call, oldest first
static int finish_search(Index* ix) {
    while (ix->pend_n > 0) KR_TRY(finish_one(ix, nullptr));
    return 0;
}

// ---- row-sharded search, exchange BEFORE the re-rank (one block of <= 1024 queries): the call is split where a shard knows its candidates' coarse scores but
// has not gathered a single fp32 row yet.  coarse half -> [the host's collective gathers every shard's k best coarse scores] -> global_theta -> re-rank half.
// The slot of the pending ring belongs to the call from the first half on; finish / finish_ex treat it like any other call once the second half is in.
static int begin_search_coarse(Index* ix, const float* q, int nq, int k, float* topk_out, hipStream_t st) {
    KR_TRY(ensure_status(ix, 1));
    while (ix->pend_n > 0 && (ix->pend[ix->pend_head].st != st || ix->pend_n == Index::PEND_MAX)) KR_TRY(finish_one(ix, nullptr));
    if (ix->pend_n > 0 && ix->pend[(ix->pend_head + ix->pend_n - 1) % Index::PEND_MAX].half)
        return fail(KR_ESTATE, "a split search is waiting for its kr_index_search_rerank_async");
    if (ix->ev_add) KR_HIP(hipStreamWaitEvent(st, ix->ev_add, 0));
    ix->force_exact = g_force_exact.load();
    ix->st.last_coarse_ms = 0.0; ix->st.last_total_ms = 0.0; ix->st.last_fine_ms = 0.0;
    Index::Pending& pd = ix->pend[(ix->pend_head + ix->pend_n) % Index::PEND_MAX];
    KR_TRY(ensure_slot_status(pd, 1));
    pd.q = q; pd.nq = nq; pd.k = k; pd.scores = nullptr; pd.rows = nullptr; pd.st = st; pd.rounds.assign(1, 0);
    pd.direct_io = false;
    pd.seq = ++ix->call_seq;
    ix->ws_owner = pd.seq;
    { int K1, cap, rmax; plan_buffers(k, K1, cap, rmax); KR_TRY(ensure_ws(ix, k, cap)); }
    KR_HIP(hipEventRecord(ix->ev[0], st));
    int rc;
    if (ix->coarse == KR_COARSE_BF16) rc = pass1_coarse<BF16>(ix, q, nq, k, 0, st, pd.rounds[0], pd.final_preset, pd.rmax, pd.byte_used);
    else rc = pass1_coarse<F16>(ix, q, nq, k, 0, st, pd.rounds[0], pd.final_preset, pd.rmax, pd.byte_used);
    if (rc) { (void)hipStreamSynchronize(st); return rc; }
    hipLaunchKernelGGL(k_local_topk, dim3(nq), dim3(256), (size_t)ix->cand_cap * sizeof(uint64_t) + 264 * sizeof(unsigned int), st, ix->cand, ix->cand_cap, ix->cnt, ix->eps, k,
                       pd.final_preset, topk_out);
    KR_HIP(hipGetLastError());
    pd.active = true; pd.half = true;
    ++ix->pend_n;
    return 0;
}

static Index::Pending* half_slot(Index* ix, hipStream_t st) {
    if (ix->pend_n == 0) return nullptr;
    Index::Pending& pd = ix->pend[(ix->pend_head + ix->pend_n - 1) % Index::PEND_MAX];
    return (pd.active && pd.half && pd.st == st && ix->ws_owner == pd.seq) ? &pd : nullptr;
}

static int search_global_theta(Index* ix, const float* gathered, int nshards, float* theta_out, hipStream_t st) {
    Index::Pending* pd = half_slot(ix, st);
    if (!pd) return fail(KR_ESTATE, "no split search is waiting on this stream (kr_index_search_coarse_async first, same stream)");
    if (nshards <= 0 || (int64_t)nshards * pd->k > MERGE_MAX) return fail(KR_EINVAL, "nshards * k = %lld exceeds %d", (long long)nshards * pd->k, MERGE_MAX);
    hipLaunchKernelGGL(k_global_theta, dim3(pd->nq), dim3(256), (size_t)nshards * pd->k * sizeof(uint64_t) + 264 * sizeof(unsigned int), st, gathered, nshards,
                       (int64_t)pd->nq * (pd->k + 1), pd->k, ix->eps, theta_out);
    KR_HIP(hipGetLastError());
    return 0;
}

static int continue_search_rerank(Index* ix, const float* theta_ext, float* scores, int64_t* rows, hipStream_t st) {
    Index::Pending* pd = half_slot(ix, st);
    if (!pd) return fail(KR_ESTATE, "no split search is waiting on this stream (kr_index_search_coarse_async first, same stream, nothing else on the handle in between)");
    pd->scores = scores; pd->rows = rows;
    const int rc = pass1_rerank(ix, pd->nq, pd->k, pd->rmax, pd->final_preset, scores, rows, 0, st, pd->status, theta_ext, pd->byte_used);
    if (rc) { (void)hipStreamSynchronize(st); pd->active = false; pd->half = false; --ix->pend_n; return rc; }
    pd->half = false;
    KR_HIP(hipEventRecord(ix->ev[3], st));
    KR_HIP(hipEventRecord(pd->ev_done, st));
    return 0;
}

// device-side final merge of per-shard top-k lists (the gathered lists are already in HBM after the RCCL all-gather): one block per query.
// Every list is sorted by (score desc, id asc) with id < 0 padding at its tail, ids are globally unique, so the merged position of entry j of
// list s is j + sum over the other lists t of the number of entries of t that precede it (binary search in LDS): no sort, no atomics on the
// output, the result is exactly kr_topk_merge's.
__global__ __launch_bounds__(256) void k_merge_lists(const float* __restrict__ scores, int64_t s_stride, const int64_t* __restrict__ ids, int64_t i_stride,
                                                     int nshards, int k, float* __restrict__ out_s, int64_t* __restrict__ out_i) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = nshards * k;
    int64_t* si = reinterpret_cast<int64_t*>(smem);                 // [n]
    float* ss = reinterpret_cast<float*>(si + n);                   // [n]
    __shared__ int valid;
    const int q = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) valid = 0;
    __syncthreads();
    int mine = 0;
    for (int s = 0; s < nshards; ++s)
        for (int j = tid; j < k; j += 256) {
            const int64_t id = ids[s * i_stride + (int64_t)q * k + j];
            si[s * k + j] = id; ss[s * k + j] = scores[s * s_stride + (int64_t)q * k + j];
            mine += id >= 0;
        }
    if (mine) atomicAdd(&valid, mine);
    __syncthreads();
    for (int e = tid; e < n; e += 256) {
        const int64_t id = si[e];
        if (id < 0) continue;
        const float sc = ss[e];
        const int s = e / k;
        int rank = e - s * k;
        for (int t = 0; t < nshards && rank < k; ++t) {
            if (t == s) continue;
            int lo = 0, hi = k;                                     // first entry of list t that does NOT precede (sc, id)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const int64_t oid = si[t * k + mid];
                const float os = ss[t * k + mid];
                const bool before = oid >= 0 && (os > sc || (os == sc && oid < id));
                if (before) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < k) { out_s[(int64_t)q * k + rank] = sc; out_i[(int64_t)q * k + rank] = id; }
    }
    for (int j = valid + tid; j < k; j += 256) { out_s[(int64_t)q * k + j] = -INFINITY; out_i[(int64_t)q * k + j] = -1; }
}

struct Scratch { Index* w = nullptr; size_t cap_x = 0, cap_q = 0, cap_i = 0, cap_s = 0, cap_r = 0; int iota_filled = 0; };
static Scratch g_scratch[64];
static std::mutex g_scratch_mu;

}  // namespace kr

using namespace kr;

// Everything the FIRST search of a (nq, k) shape would otherwise do on the spot, done now (the first KiRAG hop after a load paid 8.6 ms of
// quantisation + a 5-GB allocation at 5M rows + the workspace allocations): the search workspaces for that shape, the kernels' function attributes, and - when
// blocks of nq queries take the byte pre-scan on this index - the int8 copy (extended, if it exists, by the rows added since) with its bitmap / row list.
// The copy is re-derived rather than stored in shard files on purpose: quantising 5M rows takes 8.6 ms on the device, reading 5 GB back from disk seconds.
template <class T>
static int prepare_t(Index* ix, int nq, int k, hipStream_t st) {
    bool byte_ok = plan_block(ix, nq, k).q32 && nq <= BYTE_NQ_MAX && !ix->byte_off && g_byte_prescan.load() != 0 && ix->n >= (int64_t)g_byte_min_rows.load() && byte_dim_ok(ix->d);
    if (byte_ok) {
        if (ix->ev_add) KR_HIP(hipStreamWaitEvent(st, ix->ev_add, 0));
        byte_ok = ensure_byte_copy(ix, st);
    }
    if (byte_ok && ensure_bitmap(ix) != 0) { (void)hipGetLastError(); byte_ok = false; ix->byte_off = true; drop_byte_copy(ix); }
    const BlockPlan p = plan_block(ix, nq, k, byte_ok);
    KR_TRY(ensure_ws(ix, k, p.cap));
    KR_TRY(search_attrs(ix));
    KR_TRY(ensure_status(ix, (nq + QBLK - 1) / QBLK));
    return 0;
}

extern "C" {

int kr_index_create(int d, int metric, int coarse_dtype, int device, kr_index** out) {
    if (!out) return fail(KR_EINVAL, "out is NULL");
    *out = nullptr;
    if (metric != KR_METRIC_INNER_PRODUCT) return fail(KR_EINVAL, "only metric=inner_product (IndexFlatIP) is implemented");
    if (d < 4 || d > 4096 || (d % 4) != 0) return fail(KR_EINVAL, "vector size %d unsupported (need 4 <= d <= 4096, d %% 4 == 0)", d);
    if (coarse_dtype != KR_COARSE_BF16 && coarse_dtype != KR_COARSE_F16) return fail(KR_EINVAL, "coarse_dtype must be 0 (bf16) or 1 (f16)");
    KR_TRY(select_device(device));
    Index* ix = new Index();
    ix->d = d; ix->dpad = (int)round_up(d, 64); ix->coarse = coarse_dtype; ix->device = device;
    { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, device) == hipSuccess && p.multiProcessorCount > 0) ix->num_cu = (p.multiProcessorCount / 8) * 8; }
    hipError_t e = hipMalloc(&ix->bounds, 2 * sizeof(float));
    if (e != hipSuccess) { delete ix; return fail(KR_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
    (void)hipMemset(ix->bounds, 0, 2 * sizeof(float));
    ix->no_q32 = getenv("KIRAG_AMD_NO_Q32") != nullptr; ix->no_fine = getenv("KIRAG_AMD_NO_FINE") != nullptr;
    ix->no_mark = getenv("KIRAG_AMD_NO_MARK") != nullptr; ix->no_vmm = getenv("KIRAG_AMD_NO_VMM") != nullptr;
    ix->byte_off = getenv("KIRAG_AMD_NO_BYTE_SCAN") != nullptr;
    *out = reinterpret_cast<kr_index*>(ix);
    return 0;
}

void kr_index_destroy(kr_index* h) {
    if (!h) return;
    Index* ix = reinterpret_cast<Index*>(h);
    (void)hipSetDevice(ix->device);
    for (auto& pd : ix->pend) {
        if (pd.active) { (void)hipEventSynchronize(pd.ev_done); pd.active = false; }   // an unfinished asynchronous search: let its kernels drain
        if (pd.ev_done) (void)hipEventDestroy(pd.ev_done);
        if (pd.status) (void)hipHostFree(pd.status);
    }
    if (ix->vmm == 1) { (void)hipDeviceSynchronize(); ix->vf.release(); ix->vc.release(); ix->xf = nullptr; ix->xc = nullptr; }
    void* ptrs[] = {ix->xf, ix->xc, ix->bounds, ix->q_f, ix->q_f2, ix->theta1, ix->thr_mark, ix->bitmap, ix->rowlist, ix->q_c, ix->thr, ix->eps, ix->cnt, ix->flags, ix->cand, ix->out_s, ix->out_r,
                    ix->nrer, ix->ex_a, ix->ex_b, ix->ex_qidx, ix->blk_list, ix->blk_cnt, ix->x8, ix->sx8, ix->bounds8, ix->q8, ix->thr8, ix->mu8, ix->qmask, ix->cnt_spread};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (ix->h_status) (void)hipHostFree(ix->h_status);
    for (auto& e : ix->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ix->evc) if (e) (void)hipEventDestroy(e);
    if (ix->ev_add) (void)hipEventDestroy(ix->ev_add);
    delete ix;
}

int kr_index_reserve(kr_index* h, int64_t n_rows) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    KR_TRY(select_device(ix->device));
    if (n_rows > 0xFFFFFFF0ll) return fail(KR_EINVAL, "at most 2^32-16 rows per index shard");
    KR_TRY(finish_search(ix));
    return grow(ix, n_rows);
}

int kr_index_add(kr_index* h, const float* x, int64_t n, void* stream) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    if (n < 0 || (n > 0 && !x)) return fail(KR_EINVAL, "bad rows argument");
    if (n == 0) return 0;
    KR_TRY(select_device(ix->device));
    if (ix->n + n > 0xFFFFFFF0ll) return fail(KR_EINVAL, "at most 2^32-16 rows per index shard");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    KR_TRY(finish_search(ix));
    KR_TRY(grow(ix, ix->n + n));
    if (ix->ev_add) KR_HIP(hipStreamWaitEvent(st, ix->ev_add, 0));   // the NaN padding rows this add overwrites were written by the previous add's stream
    float* dst = ix->xf + ix->n * ix->d;
    KR_HIP(hipMemcpyAsync(dst, x, (size_t)n * ix->d * sizeof(float), hipMemcpyDefault, st));
    const unsigned grid = (unsigned)std::min<int64_t>((n + 3) / 4, (int64_t)ix->num_cu * 16);   // grid-stride: 16 blocks of 4 waves per CU
    if (ix->coarse == KR_COARSE_BF16)
        hipLaunchKernelGGL(k_add_rows<BF16>, dim3(grid), dim3(256), 0, st, dst, ix->xc + ix->n * ix->dpad, n, ix->d, ix->dpad, ix->bounds);
    else
        hipLaunchKernelGGL(k_add_rows<F16>, dim3(grid), dim3(256), 0, st, dst, ix->xc + ix->n * ix->dpad, n, ix->d, ix->dpad, ix->bounds);
    {
        const int64_t nn = ix->n + n, n_pad = round_up(nn, 256);
        if (n_pad > nn) hipLaunchKernelGGL(k_pad_nan, dim3((unsigned)(((n_pad - nn) * ix->dpad + 255) / 256)), dim3(256), 0, st, ix->xc, nn, n_pad, ix->dpad);
    }
    KR_HIP(hipGetLastError());
    // a host source must have been consumed when we return (the caller may free it); a device source is only read in stream order, and the
    // searches that follow are enqueued behind this add on the caller's stream
    if (!is_device_pointer(x)) KR_HIP(hipStreamSynchronize(st));
    else {
        if (!ix->ev_add) KR_HIP(hipEventCreateWithFlags(&ix->ev_add, hipEventDisableTiming));
        KR_HIP(hipEventRecord(ix->ev_add, st));
    }
    ix->n += n;
    return 0;
}

int64_t kr_index_ntotal(const kr_index* h) { return h ? reinterpret_cast<const Index*>(h)->n : 0; }
int kr_index_dim(const kr_index* h) { return h ? reinterpret_cast<const Index*>(h)->d : 0; }

int kr_index_get_rows(kr_index* h, int64_t start, int64_t n, float* out, void* stream) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    if (start < 0 || n < 0 || start + n > ix->n || (n > 0 && !out)) return fail(KR_EINVAL, "row range [%lld, %lld) outside [0, %lld)",
                                                                                 (long long)start, (long long)(start + n), (long long)ix->n);
    if (n == 0) return 0;
    KR_TRY(select_device(ix->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    KR_TRY(finish_search(ix));
    if (ix->ev_add) KR_HIP(hipStreamWaitEvent(st, ix->ev_add, 0));   // rows appended asynchronously on another stream
    KR_HIP(hipMemcpyAsync(out, ix->xf + start * ix->d, (size_t)n * ix->d * sizeof(float), hipMemcpyDefault, st));
    KR_HIP(hipStreamSynchronize(st));
    return 0;
}

// ---- native shard I/O (kirag_amd/retriever/index.py: ShardedIndexer.serialize / deserialize_from): the stored state of rows, exactly ----
int kr_index_coarse_dim(const kr_index* h) { return h ? reinterpret_cast<const Index*>(h)->dpad : 0; }
int kr_index_coarse_dtype(const kr_index* h) { return h ? reinterpret_cast<const Index*>(h)->coarse : 0; }

int kr_index_get_coarse(kr_index* h, int64_t start, int64_t n, uint16_t* out, void* stream) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    if (start < 0 || n < 0 || start + n > ix->n || (n > 0 && !out)) return fail(KR_EINVAL, "row range [%lld, %lld) outside [0, %lld)",
                                                                                 (long long)start, (long long)(start + n), (long long)ix->n);
    if (n == 0) return 0;
    KR_TRY(select_device(ix->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    KR_TRY(finish_search(ix));
    if (ix->ev_add) KR_HIP(hipStreamWaitEvent(st, ix->ev_add, 0));
    KR_HIP(hipMemcpyAsync(out, ix->xc + start * ix->dpad, (size_t)n * ix->dpad * 2, hipMemcpyDefault, st));
    KR_HIP(hipStreamSynchronize(st));
    return 0;
}

int kr_index_get_bounds(kr_index* h, float* out2) {
    if (!h || !out2) return fail(KR_EINVAL, "NULL argument");
    Index* ix = reinterpret_cast<Index*>(h);
    KR_TRY(select_device(ix->device));
    KR_TRY(wait_adds_host(ix));          // the bounds are updated by the add kernels; this copy runs on the NULL stream
    KR_HIP(hipMemcpy(out2, ix->bounds, 2 * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

__global__ void k_max_bounds(float* __restrict__ bounds, float b0, float b1) {
    atomicMax(reinterpret_cast<unsigned int*>(bounds), __float_as_uint(b0));
    atomicMax(reinterpret_cast<unsigned int*>(bounds) + 1, __float_as_uint(b1));
}

// append n rows whose 16-bit copy and error bounds were computed before (by kr_index_add on the index that wrote the shard): no re-quantisation
int kr_index_add_raw(kr_index* h, const float* xf, const uint16_t* xc, int64_t n, const float* bounds2, void* stream) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    if (n < 0 || (n > 0 && (!xf || !xc || !bounds2))) return fail(KR_EINVAL, "bad rows argument");
    if (!(bounds2 == nullptr || (bounds2[0] >= 0.f && bounds2[1] >= 0.f))) return fail(KR_EINVAL, "bounds must be non-negative");
    if (n == 0) return 0;
    KR_TRY(select_device(ix->device));
    if (ix->n + n > 0xFFFFFFF0ll) return fail(KR_EINVAL, "at most 2^32-16 rows per index shard");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    KR_TRY(finish_search(ix));
    KR_TRY(grow(ix, ix->n + n));
    if (ix->ev_add) KR_HIP(hipStreamWaitEvent(st, ix->ev_add, 0));
    KR_HIP(hipMemcpyAsync(ix->xf + ix->n * ix->d, xf, (size_t)n * ix->d * sizeof(float), hipMemcpyDefault, st));
    KR_HIP(hipMemcpyAsync(ix->xc + ix->n * ix->dpad, xc, (size_t)n * ix->dpad * 2, hipMemcpyDefault, st));
    hipLaunchKernelGGL(k_max_bounds, dim3(1), dim3(1), 0, st, ix->bounds, bounds2[0], bounds2[1]);
    {
        const int64_t nn = ix->n + n, n_pad = round_up(nn, 256);
        if (n_pad > nn) hipLaunchKernelGGL(k_pad_nan, dim3((unsigned)(((n_pad - nn) * ix->dpad + 255) / 256)), dim3(256), 0, st, ix->xc, nn, n_pad, ix->dpad);
    }
    KR_HIP(hipGetLastError());
    KR_HIP(hipStreamSynchronize(st));
    ix->n += n;
    return 0;
}

static int check_search_args(Index* ix, const float* q, int nq, int k, float* scores, int64_t* rows) {
    if (nq < 0 || (nq > 0 && (!q || !scores || !rows))) return fail(KR_EINVAL, "bad query/output pointers");
    if (k <= 0 || (int64_t)k > ix->n) return fail(KR_EINVAL, "k=%d must satisfy 0 < k <= ntotal=%lld", k, (long long)ix->n);
    if (k > EXACT_RC) return fail(KR_EINVAL, "k=%d exceeds the supported maximum of %d", k, EXACT_RC);
    return 0;
}

int kr_index_search(kr_index* h, const float* q, int nq, int k, float* scores, int64_t* rows, int mode, void* stream) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    KR_TRY(check_search_args(ix, q, nq, k, scores, rows));
    if (mode < 0 || mode > 2) return fail(KR_EINVAL, "mode must be 0 (auto), 1 (exact scan only) or 2 (high-precision pass only)");
    if (nq == 0) return 0;
    KR_TRY(select_device(ix->device));
    KR_TRY(finish_search(ix));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (mode == 0) {   // every block's pass 1 is enqueued before the first status word is read: one host round trip per CALL
        KR_TRY(begin_search(ix, q, nq, k, scores, rows, st));
        return finish_search(ix);
    }
    // test hooks: mode 1 = exact scan only, mode 2 = high-precision pass (+ exact scan) only; every query takes the slow path
    if (ix->ev_add) KR_HIP(hipStreamWaitEvent(st, ix->ev_add, 0));
    ix->force_exact = g_force_exact.load();
    ix->st.last_coarse_ms = 0.0; ix->st.last_total_ms = 0.0; ix->st.last_fine_ms = 0.0;
    KR_TRY(ensure_status(ix, 1));
    for (int b = 0; b < nq; b += QBLK) {
        const int nb = std::min(QBLK, nq - b);
        const std::vector<uint32_t> hflags(nb, 2u);
        int rc;
        if (ix->coarse == KR_COARSE_BF16) rc = slow_passes<BF16>(ix, q + (size_t)b * ix->d, nb, k, scores + (size_t)b * k, rows + (size_t)b * k, THETA_BLOCKS, st, hflags, false, mode != 1, true);
        else rc = slow_passes<F16>(ix, q + (size_t)b * ix->d, nb, k, scores + (size_t)b * k, rows + (size_t)b * k, THETA_BLOCKS, st, hflags, false, mode != 1, true);
        if (rc) return rc;
        ix->st.queries += nb;
    }
    return 0;
}

int kr_index_prepare(kr_index* h, int nq, int k, void* stream) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    if (nq <= 0 || k <= 0 || k > EXACT_RC) return fail(KR_EINVAL, "kr_index_prepare: nq > 0 and 0 < k <= %d", EXACT_RC);
    if (ix->n == 0) return 0;
    KR_TRY(select_device(ix->device));
    KR_TRY(finish_search(ix));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    nq = std::min(nq, QBLK); k = (int)std::min<int64_t>(k, ix->n);
    const int64_t n8_before = ix->x8 ? ix->n8 : -1;
    KR_TRY(ix->coarse == KR_COARSE_BF16 ? prepare_t<BF16>(ix, nq, k, st) : prepare_t<F16>(ix, nq, k, st));
    // ... and ONE real search of that shape on the index's own first rows (results discarded, statistics and the pre-scan's feedback state restored): whatever a
    // first search still sets up lazily - per-kernel function attributes of the stream kernels, the pending ring's pinned status records and events, the exact-score
    // paths' scratch - is then in place too (measured at 600 k rows: first search 0.45 ms against 0.20 ms steady without it)
    if (ix->n >= nq && nq <= 32 && !ix->warmed) {      // once per index (Indexer.index_data calls prepare after every append of a streamed build)
        ix->warmed = true;
        float* tmp_s = nullptr; int64_t* tmp_r = nullptr;
        if (hipMalloc(&tmp_s, (size_t)nq * k * sizeof(float)) == hipSuccess && hipMalloc(&tmp_r, (size_t)nq * k * sizeof(int64_t)) == hipSuccess) {
            const kr_search_stats saved = ix->st;
            const int bad = ix->byte_bad, pause = ix->byte_pause;
            int rc = begin_search(ix, ix->xf, nq, k, tmp_s, tmp_r, st);      // queries = rows 0 .. nq-1 (device pointers: in-place, nothing is staged)
            if (rc == 0) rc = finish_search(ix);
            ix->st = saved; ix->byte_bad = bad; ix->byte_pause = pause;
            if (rc != 0) { (void)hipFree(tmp_s); (void)hipFree(tmp_r); return rc; }
        } else (void)hipGetLastError();
        if (tmp_s) (void)hipFree(tmp_s);
        if (tmp_r) (void)hipFree(tmp_r);
    }
    if ((ix->x8 ? ix->n8 : -1) != n8_before) {
        // the copy was built / extended by kernels on `st`: searches on ANOTHER stream must not read it early - they already wait for ev_add (the event behind
        // the last asynchronous add), so it is recorded again behind these kernels
        if (!ix->ev_add) KR_HIP(hipEventCreateWithFlags(&ix->ev_add, hipEventDisableTiming));
        KR_HIP(hipEventRecord(ix->ev_add, st));
    }
    return 0;
}

int kr_index_search_async(kr_index* h, const float* q, int nq, int k, float* scores, int64_t* rows, void* stream) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    KR_TRY(check_search_args(ix, q, nq, k, scores, rows));
    if (nq == 0) return 0;
    KR_TRY(select_device(ix->device));
    return begin_search(ix, q, nq, k, scores, rows, reinterpret_cast<hipStream_t>(stream));   // up to PEND_MAX calls outstanding per handle (one stream)
}

int kr_index_search_coarse_async(kr_index* h, const float* q, int nq, int k, float* topk, void* stream) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    if (!q || !topk || nq <= 0 || nq > QBLK) return fail(KR_EINVAL, "kr_index_search_coarse_async: 0 < nq <= %d queries, non-NULL pointers", QBLK);
    if (k <= 0 || (int64_t)k > ix->n || k > EXACT_RC) return fail(KR_EINVAL, "k=%d must satisfy 0 < k <= min(ntotal=%lld, %d)", k, (long long)ix->n, EXACT_RC);
    if (!is_device_pointer(topk)) return fail(KR_EINVAL, "kr_index_search_coarse_async: topk must be device memory (it feeds the collective)");
    KR_TRY(select_device(ix->device));
    return begin_search_coarse(ix, q, nq, k, topk, reinterpret_cast<hipStream_t>(stream));
}

int kr_index_search_global_theta(kr_index* h, const float* gathered, int nshards, float* theta, void* stream) {
    if (!h || !gathered || !theta) return fail(KR_EINVAL, "NULL argument");
    Index* ix = reinterpret_cast<Index*>(h);
    KR_TRY(select_device(ix->device));
    return search_global_theta(ix, gathered, nshards, theta, reinterpret_cast<hipStream_t>(stream));
}

int kr_index_search_rerank_async(kr_index* h, const float* theta, float* scores, int64_t* rows, void* stream) {
    if (!h || !scores || !rows) return fail(KR_EINVAL, "NULL argument");
    Index* ix = reinterpret_cast<Index*>(h);
    KR_TRY(select_device(ix->device));
    return continue_search_rerank(ix, theta, scores, rows, reinterpret_cast<hipStream_t>(stream));
}

int kr_index_search_finish(kr_index* h) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    KR_TRY(select_device(ix->device));
    return finish_search(ix);
}

int kr_index_search_finish_ex(kr_index* h, int64_t* flagged, int cap, int* ncalls) {
    if (!h || !ncalls || (cap > 0 && !flagged) || cap < 0) return fail(KR_EINVAL, "bad arguments");
    Index* ix = reinterpret_cast<Index*>(h);
    KR_TRY(select_device(ix->device));
    *ncalls = 0;
    while (ix->pend_n > 0) {
        int64_t fl = 0;
        KR_TRY(finish_one(ix, &fl));
        if (*ncalls < cap) flagged[*ncalls] = fl;
        ++*ncalls;
    }
    return 0;
}

int kr_index_search_finish_one(kr_index* h, int64_t* flagged) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    KR_TRY(select_device(ix->device));
    int64_t fl = 0;
    KR_TRY(finish_one(ix, &fl));
    if (flagged) *flagged = fl;
    return 0;
}

int kr_index_search_pending(const kr_index* h) { return h ? reinterpret_cast<const Index*>(h)->pend_n : 0; }

// exact top-k of q x^T for a small, transient candidate set (the KiRAG loop's aligner step): canonical scores of every (query, row)
// pair by k_exact_scan + the sort tree, on a per-device scratch Index (no 16-bit copy, no certificate needed: this IS the exact scan)
int kr_score_topk(const float* q, int nq, const float* x, int64_t n, int d, int k, float* scores, int64_t* rows, int device, void* stream) {
    if (!q || !x || !scores || !rows) return fail(KR_EINVAL, "NULL argument");
    if (d < 4 || d > 4096 || (d % 4) != 0) return fail(KR_EINVAL, "vector size %d unsupported (need 4 <= d <= 4096, d %% 4 == 0)", d);
    if (nq <= 0 || nq > 65535) return fail(KR_EINVAL, "nq=%d must satisfy 0 < nq <= 65535", nq);
    if (n <= 0 || n > 0xFFFFFFF0ll) return fail(KR_EINVAL, "bad row count");
    if (k <= 0 || (int64_t)k > n || k > EXACT_RC) return fail(KR_EINVAL, "k=%d must satisfy 0 < k <= min(n=%lld, %d)", k, (long long)n, EXACT_RC);
    KR_TRY(select_device(device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (device < 0 || device >= 64) return fail(KR_EINVAL, "device %d out of range", device);
    // per-device scratch shared by all host threads (calls are serialised by the mutex; the buffers persist between calls and are
    // freed by kr_release_scratch())
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    Scratch& sc = g_scratch[device];
    if (!sc.w) { sc.w = new Index(); sc.w->device = device; }
    Index* w = sc.w;
    w->force_exact = g_force_exact.load();
    auto regrow = [](auto*& p, size_t& have, size_t need) -> int {
        if (need <= have) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; have = 0;
        KR_HIP(hipMalloc(reinterpret_cast<void**>(&p), need));
        have = need;
        return 0;
    };
    KR_TRY(regrow(w->xf, sc.cap_x, (size_t)n * d * sizeof(float)));
    KR_TRY(regrow(w->q_f, sc.cap_q, (size_t)nq * d * sizeof(float)));
    KR_TRY(regrow(w->ex_qidx, sc.cap_i, (size_t)nq * sizeof(int)));
    KR_TRY(regrow(w->out_s, sc.cap_s, (size_t)nq * k * sizeof(float)));
    KR_TRY(regrow(w->out_r, sc.cap_r, (size_t)nq * k * sizeof(int64_t)));
    w->d = d; w->n = n;
    KR_HIP(hipMemcpyAsync(w->xf, x, (size_t)n * d * sizeof(float), hipMemcpyDefault, st));
    KR_HIP(hipMemcpyAsync(w->q_f, q, (size_t)nq * d * sizeof(float), hipMemcpyDefault, st));
    if (sc.cap_i != (size_t)sc.iota_filled * sizeof(int) || sc.iota_filled < nq) {   // ex_qidx[i] = i is written once per (re)allocation, not per call
        const int cnt = (int)(sc.cap_i / sizeof(int));
        std::vector<int> iota(cnt);
        for (int i = 0; i < cnt; ++i) iota[i] = i;
        KR_HIP(hipMemcpyAsync(w->ex_qidx, iota.data(), (size_t)cnt * sizeof(int), hipMemcpyHostToDevice, st));
        KR_HIP(hipStreamSynchronize(st));   // iota is a heap host buffer
        sc.iota_filled = cnt;
    }
    KR_TRY(exact_scan(w, nq, k, st));
    KR_HIP(hipMemcpyAsync(scores, w->out_s, (size_t)nq * k * sizeof(float), hipMemcpyDefault, st));
    KR_HIP(hipMemcpyAsync(rows, w->out_r, (size_t)nq * k * sizeof(int64_t), hipMemcpyDefault, st));
    KR_HIP(hipStreamSynchronize(st));
    return 0;
}

#ifdef KR_STAMP
// diagnostic build only: read and clear the per-wave stamp sums of gemm_nt_pingpong (this translation unit's copy: the coarse scan)
int kr_debug_read_stamps(unsigned long long* out256) {
    if (hipMemcpyFromSymbol(out256, HIP_SYMBOL(kr_stamp_buf), 256 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[256] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(kr_stamp_buf), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

void kr_release_scratch(void) {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    for (int dev = 0; dev < 64; ++dev) {
        Scratch& sc = g_scratch[dev];
        if (!sc.w) continue;
        if (hipSetDevice(dev) == hipSuccess) {
            void* ptrs[] = {sc.w->xf, sc.w->q_f, sc.w->ex_qidx, sc.w->out_s, sc.w->out_r, sc.w->ex_a, sc.w->ex_b};
            for (void* p : ptrs) if (p) (void)hipFree(p);
        }
        delete sc.w;
        sc = Scratch{};
    }
}

int kr_index_stats(kr_index* h, kr_search_stats* out, int reset) {
    if (!h || !out) return fail(KR_EINVAL, "NULL argument");
    Index* ix = reinterpret_cast<Index*>(h);
    *out = ix->st;
    out->va_retired_bytes = (int64_t)va_retired_total();
    out->grow_mode = ix->vmm;
    out->byte_rows = ix->x8 ? ix->n8 : 0;
    if (reset) ix->st = kr_search_stats{};
    return 0;
}

int kr_topk_merge(const float* scores, const int64_t* ids, int nshards, int nq, int k, float* out_scores, int64_t* out_ids) {
    if (!scores || !ids || !out_scores || !out_ids || nshards <= 0 || nq < 0 || k <= 0) return fail(KR_EINVAL, "bad merge arguments");
    // every shard list is already sorted by (score desc, id asc): k-way merge by repeated head selection.  Queries are independent: large merges
    // (8 shards x 1000 queries x 100 = 0.75 ms on one core, on the critical path of every multi-GPU step) are split over up to 8 host threads.
    auto merge_range = [=](int q_begin, int q_end) {
        std::vector<int> head(nshards);
        for (int q = q_begin; q < q_end; ++q) {
            std::fill(head.begin(), head.end(), 0);
            for (int j = 0; j < k; ++j) {
                int best = -1; float bs = 0.f; int64_t bi = 0;
                for (int s = 0; s < nshards; ++s) {
                    if (head[s] >= k) continue;
                    const size_t o = ((size_t)s * nq + q) * k + head[s];
                    const float sc = scores[o]; const int64_t id = ids[o];
                    if (id < 0) { head[s] = k; continue; }   // shard shorter than k (padding)
                    if (best < 0 || sc > bs || (sc == bs && id < bi)) { best = s; bs = sc; bi = id; }
                }
                if (best < 0) { out_scores[(size_t)q * k + j] = -INFINITY; out_ids[(size_t)q * k + j] = -1; continue; }
                out_scores[(size_t)q * k + j] = bs; out_ids[(size_t)q * k + j] = bi; head[best]++;
            }
        }
    };
    const int64_t work = (int64_t)nq * k * nshards;
    int nthreads = (int)std::min<int64_t>(std::min<int64_t>(8, std::max(1u, std::thread::hardware_concurrency())), work / 65536);
    nthreads = std::min(nthreads, nq);
    if (nthreads <= 1) { merge_range(0, nq); return 0; }
    std::vector<std::thread> pool;
    pool.reserve(nthreads - 1);
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(merge_range, (int)((int64_t)nq * t / nthreads), (int)((int64_t)nq * (t + 1) / nthreads));
    merge_range(0, nq / nthreads);
    for (auto& th : pool) th.join();
    return 0;
}

int kr_topk_merge_device(const float* scores, int64_t score_shard_stride, const int64_t* ids, int64_t id_shard_stride, int nshards, int nq, int k,
                         float* out_scores, int64_t* out_ids, int device, void* stream) {
    if (!scores || !ids || !out_scores || !out_ids || nshards <= 0 || nq < 0 || k <= 0) return fail(KR_EINVAL, "bad merge arguments");
    if ((int64_t)nshards * k > MERGE_MAX) return fail(KR_EINVAL, "nshards * k = %lld exceeds %d (use kr_topk_merge)", (long long)nshards * k, MERGE_MAX);
    if (device < 0 || device >= 64) return fail(KR_EINVAL, "device %d out of range", device);
    if (nq == 0) return 0;
    KR_TRY(select_device(device));
    static DevOnce once;
    KR_TRY(once_per_device(once, device, [&]() -> int {
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_merge_lists), hipFuncAttributeMaxDynamicSharedMemorySize, MERGE_MAX * 12));
        return 0;
    }));
    hipLaunchKernelGGL(k_merge_lists, dim3((unsigned)nq), dim3(256), (size_t)nshards * k * 12, reinterpret_cast<hipStream_t>(stream), scores, score_shard_stride,
                       ids, id_shard_stride, nshards, k, out_scores, out_ids);
    KR_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
