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
//   4. flagged queries are re-answered by the exact full scan (k_exact_scan + k_sort_chunks tree).
// The result is therefore ALWAYS the exact top-k under the canonical score; the 16-bit scan only decides
// which rows get the fp64 treatment.
#include "gemm_nt.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace kr {

using ShapeC = GemmShape<128, 128, 2, 2>;   // coarse scan tile: 128 corpus rows x 128 queries
constexpr int QBLK = 1024;                   // queries per search block (reference index_batch_size)
constexpr int EXACT_RC = 1024;               // rows per block of the exact scan
constexpr int SORT_CHUNK = 4096;             // keys per block of the merge tree

struct Index {
    int d = 0, dpad = 0, coarse = KR_COARSE_BF16, device = 0;
    int64_t n = 0, cap_rows = 0;
    float* xf = nullptr;       // [cap_rows, d] fp32 master
    uint16_t* xc = nullptr;    // [cap_rows, dpad] 16-bit coarse copy
    float* bounds = nullptr;   // [2] device: max_x ||x - c(x)||_2 , max_x ||c(x)||_2   (non-negative -> uint order)
    // search workspace (sized for QBLK queries)
    float* q_f = nullptr;      // [QBLK, d]
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
    int out_k = 0;
    uint64_t* ex_a = nullptr; uint64_t* ex_b = nullptr; size_t ex_bytes = 0;  // exact-scan ping/pong
    int* ex_qidx = nullptr;    // [QBLK] flagged query list
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t evc[2 * 16] = {};   // begin/end pairs around each coarse round (roofline timing)
    kr_search_stats st{};
};

// ---------------------------------------------------------------------------------------------------------
// add: fp32 rows -> 16-bit copy + quantisation-error bounds
// ---------------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void k_add_rows(const float* __restrict__ xf, uint16_t* __restrict__ xc, int64_t n, int d, int dpad,
                                                  float* __restrict__ bounds) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
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
    if (lane == 0) {
        // slack factor covers the fp32 rounding of these sums themselves; NaN rows (never retrievable) are skipped
        const float e = sqrtf(e2) * 1.0001f, c = sqrtf(c2) * 1.0001f;
        if (e == e) atomicMax(reinterpret_cast<unsigned int*>(bounds), __float_as_uint(e));
        if (c == c) atomicMax(reinterpret_cast<unsigned int*>(bounds) + 1, __float_as_uint(c));
    }
}

// queries: fp32 -> 16-bit (rows >= nq zero), eps_q, and state reset
template <class T>
__global__ __launch_bounds__(64) void k_prep_queries(const float* __restrict__ qf, uint16_t* __restrict__ qc, int nq, int d, int dpad,
                                                     const float* __restrict__ bounds, float* __restrict__ eps, float* __restrict__ thr,
                                                     uint32_t* __restrict__ cnt, uint32_t* __restrict__ flags) {
    const int q = blockIdx.x, lane = threadIdx.x;
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
    int64_t tile_begin, tile_count;   // this round covers permuted tile slots [tile_begin, tile_begin + tile_count)
    int64_t ntiles, perm_mul;         // slot -> tile = (slot * perm_mul) % ntiles   (perm_mul coprime to ntiles)
};

template <class T>
__global__ __launch_bounds__(ShapeC::NTHREADS, 2) void k_coarse(CoarseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t tslot; int tn;
    if (!xcd_tile_map(blockIdx.x, a.tile_count, a.nq_pad / ShapeC::BN, tslot, tn)) return;
    // ntiles < 2^25 (2^32 rows / 128) so the product stays below 2^50
    const int64_t tile = (int64_t)(((uint64_t)(a.tile_begin + tslot) * (uint64_t)a.perm_mul) % (uint64_t)a.ntiles);
    const int64_t m0 = tile * ShapeC::BM;
    const int n0 = tn * ShapeC::BN;
    gemm_nt_block<T, ShapeC>(a.xc, a.dpad, a.n, a.qc, a.dpad, a.nq_pad, a.dpad, m0, n0, smem, [&](AccTile<ShapeC>& acc) {
#pragma unroll
        for (int ni = 0; ni < ShapeC::TN; ++ni) {
            const int q = n0 + acc.col(ni);
            const float t = a.thr[q];
#pragma unroll
            for (int mi = 0; mi < ShapeC::TM; ++mi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s = acc.v[mi][ni][r];
                    if (s >= t) {
                        const int64_t row = m0 + acc.row(mi, r);
                        if (row < a.n) {
                            const uint32_t slot = atomicAdd(&a.cnt[q], 1u);
                            if (slot < (uint32_t)a.cand_cap) a.cand[(int64_t)q * a.cand_cap + slot] = make_key(s, (uint32_t)row);
                        }
                    }
                }
            }
        }
    });
}

// sort a query's candidate buffer (descending); keep `keep` entries (or all when keep < 0); publish the new threshold
__global__ __launch_bounds__(256) void k_select(uint64_t* __restrict__ cand, int cand_cap, uint32_t* __restrict__ cnt,
                                                uint32_t* __restrict__ flags, float* __restrict__ thr, int keep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem);
    const int q = blockIdx.x, tid = threadIdx.x;
    uint32_t m = cnt[q];
    if (m > (uint32_t)cand_cap) {   // entries beyond the capacity were dropped: the query goes to the exact scan
        if (tid == 0) flags[q] |= 1u;
        m = (uint32_t)cand_cap;
    }
    if (m == 0) return;
    int P = 1; while (P < (int)m) P <<= 1;
    uint64_t* c = cand + (int64_t)q * cand_cap;
    for (int i = tid; i < P; i += 256) s[i] = (i < (int)m) ? c[i] : 0ull;   // key 0 sorts last
    bitonic_sort_desc(s, P, tid, 256);
    const int out = (keep >= 0 && keep < (int)m) ? keep : (int)m;
    for (int i = tid; i < out; i += 256) c[i] = s[i];
    if (tid == 0) {
        cnt[q] = (uint32_t)out;
        if (keep >= 0 && (int)m >= keep) thr[q] = key_score(s[keep - 1]);
    }
}

// exactness certificate + fp64 re-rank of the buffer prefix {coarse >= b_k - 2 eps}
__global__ __launch_bounds__(256) void k_rerank(const uint64_t* __restrict__ cand, int cand_cap, const uint32_t* __restrict__ cnt,
                                                uint32_t* __restrict__ flags, const float* __restrict__ thr, const float* __restrict__ eps,
                                                const float* __restrict__ qf, const float* __restrict__ xf, int d, int k,
                                                float* __restrict__ out_s, int64_t* __restrict__ out_r, uint32_t* __restrict__ nrer) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* s = reinterpret_cast<uint64_t*>(smem);
    int& r_sh = *reinterpret_cast<int*>(smem + (size_t)cand_cap * sizeof(uint64_t));   // tail word of the dynamic region (no static LDS: G17)
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = (int)cnt[q];
    const uint64_t* c = cand + (int64_t)q * cand_cap;
    if (m < k) {   // fewer emittable rows than k (NaN rows, overflow truncation): exact scan decides
        if (tid == 0) { flags[q] |= 2u; nrer[q] = 0; }
        return;
    }
    const float bk = key_score(c[k - 1]);
    const float theta = bk - 2.f * eps[q];
    // the buffer is complete for coarse scores strictly above thr (ties AT thr may have been compacted away)
    const bool certified = (flags[q] == 0u) && (theta > thr[q]);
    if (tid == 0) r_sh = 0;
    __syncthreads();
    int local = 0;
    for (int i = tid; i < m; i += 256) local += (key_score(c[i]) >= theta) ? 1 : 0;   // sorted desc: this is a prefix
    atomicAdd(&r_sh, local);
    __syncthreads();
    const int r = r_sh;
    if (tid == 0) { nrer[q] = (uint32_t)r; if (!certified) flags[q] |= 2u; }
    int P = 1; while (P < r) P <<= 1;
    const float* qv = qf + (int64_t)q * d;
    for (int i = wave; i < P; i += 4) {
        uint64_t key = 0ull;
        if (i < r) {
            const uint32_t row = key_row(c[i]);
            const double e = canonical_dot_wave(qv, xf + (int64_t)row * d, d, lane);
            key = make_key((float)e, row);
        }
        if (lane == 0) s[i] = key;
    }
    bitonic_sort_desc(s, P, tid, 256);
    for (int j = tid; j < k; j += 256) {
        out_s[(int64_t)q * k + j] = key_score(s[j]);
        out_r[(int64_t)q * k + j] = (int64_t)key_row(s[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// exact full scan (fallback for flagged queries, and mode = 1)
// ---------------------------------------------------------------------------------------------------------
// grid (nchunks, nqf): canonical score of every row of the chunk for one query, chunk-local top-kk keys
__global__ __launch_bounds__(256) void k_exact_scan(const float* __restrict__ qf, const int* __restrict__ qidx, const float* __restrict__ xf,
                                                    int64_t n, int d, int kk, uint64_t* __restrict__ out, int nchunks) {
    __shared__ uint64_t s[EXACT_RC];
    const int chunk = blockIdx.x, qi = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* qv = qf + (int64_t)qidx[qi] * d;
    const int64_t r0 = (int64_t)chunk * EXACT_RC;
    for (int i = wave; i < EXACT_RC; i += 4) {
        const int64_t row = r0 + i;
        uint64_t key = 0ull;
        if (row < n) {
            const double e = canonical_dot_wave(qv, xf + row * d, d, lane);
            key = make_key((float)e, (uint32_t)row);
            if (key == 0ull) key = 1ull;   // cannot happen for row < 2^32-1; keeps "0 = padding" unambiguous
        }
        if (lane == 0) s[i] = key;
    }
    bitonic_sort_desc(s, EXACT_RC, tid, 256);
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
static int grow(Index* ix, int64_t want) {
    if (want <= ix->cap_rows) return 0;
    int64_t ncap = std::max<int64_t>(want, ix->cap_rows + ix->cap_rows / 2);
    ncap = round_up(ncap, ShapeC::BM);
    float* nf = nullptr; uint16_t* nc = nullptr;
    KR_HIP(hipMalloc(&nf, (size_t)ncap * ix->d * sizeof(float)));
    hipError_t e = hipMalloc(&nc, (size_t)ncap * ix->dpad * 2);
    if (e != hipSuccess) { (void)hipFree(nf); return fail(KR_ENOMEM, "hipMalloc of the coarse copy failed: %s", hipGetErrorString(e)); }
    if (ix->n > 0) {
        KR_HIP(hipMemcpy(nf, ix->xf, (size_t)ix->n * ix->d * sizeof(float), hipMemcpyDeviceToDevice));
        KR_HIP(hipMemcpy(nc, ix->xc, (size_t)ix->n * ix->dpad * 2, hipMemcpyDeviceToDevice));
    }
    if (ix->xf) (void)hipFree(ix->xf);
    if (ix->xc) (void)hipFree(ix->xc);
    ix->xf = nf; ix->xc = nc; ix->cap_rows = ncap;
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
        KR_HIP(hipMalloc(&ix->ex_qidx, QBLK * sizeof(int)));
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
    const int64_t nchunks = (ix->n + EXACT_RC - 1) / EXACT_RC;
    const int kk = std::min<int>(k, EXACT_RC);
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
        hipLaunchKernelGGL(k_exact_scan, dim3((unsigned)nchunks, g), dim3(256), 0, st, ix->q_f, ix->ex_qidx + g0, ix->xf, ix->n, ix->d, kk,
                           ix->ex_a, (int)nchunks);
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

template <class T>
static int search_block(Index* ix, const float* q, int nq, int k, float* scores, int64_t* rows, int mode, hipStream_t st) {
    const int nq_pad = (int)round_up(nq, ShapeC::BN);
    // over-fetch K1 and buffer capacity: K1 = max(64, pow2 >= 2.5 k); cap = 16 K1; growth 8x per round
    int K1 = std::max(64, next_pow2((5 * k + 1) / 2));
    int cap = 16 * K1;
    const bool fast_ok = (cap <= 8192) && mode == 0;
    if (!fast_ok) { K1 = 64; cap = 1024; }
    KR_TRY(ensure_ws(ix, k, cap));
    KR_HIP(hipMemcpyAsync(ix->q_f, q, (size_t)nq * ix->d * sizeof(float), hipMemcpyDefault, st));
    KR_HIP(hipEventRecord(ix->ev[0], st));
    std::vector<uint32_t> hflags(nq, 2u);
    double coarse_ms = 0.0;
    if (fast_ok) {
        hipLaunchKernelGGL(k_prep_queries<T>, dim3(nq_pad), dim3(64), 0, st, ix->q_f, ix->q_c, nq, ix->d, ix->dpad, ix->bounds, ix->eps, ix->thr,
                           ix->cnt, ix->flags);
        CoarseArgs a;
        a.xc = ix->xc; a.n = ix->n; a.dpad = ix->dpad; a.qc = ix->q_c; a.nq_pad = nq_pad;
        a.thr = ix->thr; a.cnt = ix->cnt; a.flags = ix->flags; a.cand = ix->cand; a.cand_cap = ix->cand_cap;
        a.ntiles = (ix->n + ShapeC::BM - 1) / ShapeC::BM;
        // interleaving permutation: multiplier near ntiles / golden ratio, coprime to ntiles
        int64_t mul = std::max<int64_t>(1, (int64_t)((double)a.ntiles * 0.6180339887498949));
        while (gcd64(mul, a.ntiles) != 1) ++mul;
        a.perm_mul = mul % a.ntiles; if (a.perm_mul == 0) a.perm_mul = 1;
        const int tn_count = nq_pad / ShapeC::BN;
        int64_t done = 0;
        int64_t step = std::max<int64_t>(1, cap / ShapeC::BM);   // round 0: at most `cap` rows -> cannot overflow
        int round = 0;
        while (done < a.ntiles) {
            const int64_t cnt_t = std::min<int64_t>(step, a.ntiles - done);
            a.tile_begin = done; a.tile_count = cnt_t;
            const int64_t grid = round_up(cnt_t, 8) * tn_count;
            if (round < 16) KR_HIP(hipEventRecord(ix->evc[2 * round], st));
            hipLaunchKernelGGL(k_coarse<T>, dim3((unsigned)grid), dim3(ShapeC::NTHREADS), ShapeC::LDS_BYTES, st, a);
            if (round < 16) KR_HIP(hipEventRecord(ix->evc[2 * round + 1], st));
            ++round;
            done += cnt_t;
            const bool last = (done >= a.ntiles);
            // after the last round the whole buffer is sorted and kept; thr is left untouched
            hipLaunchKernelGGL(k_select, dim3(nq), dim3(256), (size_t)ix->cand_cap * sizeof(uint64_t), st, ix->cand, ix->cand_cap, ix->cnt,
                               ix->flags, ix->thr, last ? -1 : K1);
            step = done * 7;   // next round: 7x the rows seen so far (expected 7*K1 new candidates + K1 kept = cap/2)
            ix->st.coarse_rounds++;
        }
        hipLaunchKernelGGL(k_rerank, dim3(nq), dim3(256), (size_t)ix->cand_cap * sizeof(uint64_t) + 16, st, ix->cand, ix->cand_cap, ix->cnt, ix->flags,
                           ix->thr, ix->eps, ix->q_f, ix->xf, ix->d, k, ix->out_s, ix->out_r, ix->nrer);
        KR_HIP(hipGetLastError());
        KR_HIP(hipMemcpyAsync(hflags.data(), ix->flags, nq * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        KR_HIP(hipStreamSynchronize(st));
        for (int r = 0; r < round && r < 16; ++r) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ix->evc[2 * r], ix->evc[2 * r + 1]) == hipSuccess) coarse_ms += ms;
        }
        std::vector<uint32_t> hn(nq);
        KR_HIP(hipMemcpy(hn.data(), ix->nrer, nq * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int i = 0; i < nq; ++i) ix->st.reranked_rows += hn[i];
    }
    std::vector<int> fl;
    for (int i = 0; i < nq; ++i) {
        if (hflags[i]) { fl.push_back(i); if (hflags[i] & 1u) ix->st.overflow++; }
    }
    if (!fl.empty()) {
        KR_HIP(hipMemcpyAsync(ix->ex_qidx, fl.data(), fl.size() * sizeof(int), hipMemcpyHostToDevice, st));
        KR_TRY(exact_scan(ix, (int)fl.size(), k, st));
    }
    KR_HIP(hipMemcpyAsync(scores, ix->out_s, (size_t)nq * k * sizeof(float), hipMemcpyDefault, st));
    KR_HIP(hipMemcpyAsync(rows, ix->out_r, (size_t)nq * k * sizeof(int64_t), hipMemcpyDefault, st));
    KR_HIP(hipEventRecord(ix->ev[3], st));
    KR_HIP(hipStreamSynchronize(st));
    float tot = 0.f;
    if (hipEventElapsedTime(&tot, ix->ev[0], ix->ev[3]) == hipSuccess) ix->st.last_total_ms += tot;
    ix->st.last_coarse_ms += coarse_ms;
    ix->st.queries += nq;
    ix->st.fallback += (int64_t)fl.size();
    ix->st.certified += nq - (int64_t)fl.size();
    return 0;
}

}  // namespace kr

using namespace kr;

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
    hipError_t e = hipMalloc(&ix->bounds, 2 * sizeof(float));
    if (e != hipSuccess) { delete ix; return fail(KR_ENOMEM, "hipMalloc failed: %s", hipGetErrorString(e)); }
    (void)hipMemset(ix->bounds, 0, 2 * sizeof(float));
    *out = reinterpret_cast<kr_index*>(ix);
    return 0;
}

void kr_index_destroy(kr_index* h) {
    if (!h) return;
    Index* ix = reinterpret_cast<Index*>(h);
    (void)hipSetDevice(ix->device);
    void* ptrs[] = {ix->xf, ix->xc, ix->bounds, ix->q_f, ix->q_c, ix->thr, ix->eps, ix->cnt, ix->flags, ix->cand, ix->out_s, ix->out_r,
                    ix->nrer, ix->ex_a, ix->ex_b, ix->ex_qidx};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& e : ix->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ix->evc) if (e) (void)hipEventDestroy(e);
    delete ix;
}

int kr_index_reserve(kr_index* h, int64_t n_rows) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    KR_TRY(select_device(ix->device));
    if (n_rows > 0xFFFFFFF0ll) return fail(KR_EINVAL, "at most 2^32-16 rows per index shard");
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
    KR_TRY(grow(ix, ix->n + n));
    float* dst = ix->xf + ix->n * ix->d;
    KR_HIP(hipMemcpyAsync(dst, x, (size_t)n * ix->d * sizeof(float), hipMemcpyDefault, st));
    const unsigned grid = (unsigned)((n + 3) / 4);
    if (ix->coarse == KR_COARSE_BF16)
        hipLaunchKernelGGL(k_add_rows<BF16>, dim3(grid), dim3(256), 0, st, dst, ix->xc + ix->n * ix->dpad, n, ix->d, ix->dpad, ix->bounds);
    else
        hipLaunchKernelGGL(k_add_rows<F16>, dim3(grid), dim3(256), 0, st, dst, ix->xc + ix->n * ix->dpad, n, ix->d, ix->dpad, ix->bounds);
    KR_HIP(hipGetLastError());
    KR_HIP(hipStreamSynchronize(st));   // x may be a pageable host buffer the caller frees on return
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
    KR_HIP(hipMemcpyAsync(out, ix->xf + start * ix->d, (size_t)n * ix->d * sizeof(float), hipMemcpyDefault, st));
    KR_HIP(hipStreamSynchronize(st));
    return 0;
}

int kr_index_search(kr_index* h, const float* q, int nq, int k, float* scores, int64_t* rows, int mode, void* stream) {
    if (!h) return fail(KR_EINVAL, "index is NULL");
    Index* ix = reinterpret_cast<Index*>(h);
    if (nq < 0 || (nq > 0 && (!q || !scores || !rows))) return fail(KR_EINVAL, "bad query/output pointers");
    if (k <= 0 || (int64_t)k > ix->n) return fail(KR_EINVAL, "k=%d must satisfy 0 < k <= ntotal=%lld", k, (long long)ix->n);
    if (k > EXACT_RC) return fail(KR_EINVAL, "k=%d exceeds the supported maximum of %d", k, EXACT_RC);
    if (mode != 0 && mode != 1) return fail(KR_EINVAL, "mode must be 0 (auto) or 1 (exact scan)");
    if (nq == 0) return 0;
    KR_TRY(select_device(ix->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ix->st.last_coarse_ms = 0.0; ix->st.last_total_ms = 0.0;
    for (int b = 0; b < nq; b += QBLK) {
        const int nb = std::min(QBLK, nq - b);
        int rc;
        if (ix->coarse == KR_COARSE_BF16) rc = search_block<BF16>(ix, q + (size_t)b * ix->d, nb, k, scores + (size_t)b * k, rows + (size_t)b * k, mode, st);
        else rc = search_block<F16>(ix, q + (size_t)b * ix->d, nb, k, scores + (size_t)b * k, rows + (size_t)b * k, mode, st);
        if (rc) return rc;
    }
    return 0;
}

int kr_index_stats(kr_index* h, kr_search_stats* out, int reset) {
    if (!h || !out) return fail(KR_EINVAL, "NULL argument");
    Index* ix = reinterpret_cast<Index*>(h);
    *out = ix->st;
    if (reset) ix->st = kr_search_stats{};
    return 0;
}

int kr_topk_merge(const float* scores, const int64_t* ids, int nshards, int nq, int k, float* out_scores, int64_t* out_ids) {
    if (!scores || !ids || !out_scores || !out_ids || nshards <= 0 || nq < 0 || k <= 0) return fail(KR_EINVAL, "bad merge arguments");
    // every shard list is already sorted by (score desc, id asc): k-way merge by repeated head selection
    std::vector<int> head(nshards);
    for (int q = 0; q < nq; ++q) {
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
    return 0;
}

}  // extern "C"
