// BERT-family sentence encoder forward on MI355X — replaces HF BertModel.forward + pooling + F.normalize behind
// retriever/encoders.py (E5Encoder.forward :67-77, BGEEncoder.forward :106-118; retriever/e5.py:51-61).
//
// Layout.  The [B,S] batch is PACKED on the device: only positions with attention_mask != 0 become token rows
// (sequence b owns rows [off_b, off_b + nq_b), off_b % 4 == 0; absolute position ids are kept per token), so the
// projections run on sum(len) rows instead of B*S.  No host round trip: grids are sized for B*(S+4) rows and blocks
// beyond the device-side total exit.  Residual stream = xb, 16-bit (it is also the MFMA operand) + by default xlo, ONE byte per element holding the
// remainder in units of ulp(hi) / 256 (lo_encode: 19 significand bits with f16 operands).  Operand type (f16 default, bf16) and the low half are fixed
// per handle at creation (encoder_api.hip: kr_encoder_create_ex); why the defaults are what they are: DESIGN.md sections 2 and 4.2a (golden set G10:
// with outlier hidden channels two orders above the median (out3) only f16 + low half stays inside the 1e-3 score tolerance with margin: 1.4e-4; on the 5 x harsher
// out16 set it measures 1.0e-3 and its test bar is 1.5e-3).  MFMA operands
// 16-bit (xb, q, k, vT, ctx, h), fp32 accumulation everywhere.
//
// Per layer (post-LN BERT):  ONE GEMM [Wq/8|Wk|Wv] x -> q, k (row-major) and v TRANSPOSED [H, T] (so that attention reads
// V^T fragments contiguously); attention = one block per (sequence, head group) with K and V^T staged once in LDS, swapped QK^T so
// the softmax reductions are in-lane, P^T fed from the accumulator straight into the V^T.P^T MFMA;  Wo ctx + b +
// residual -> LayerNorm;  W1 x + b -> erf-GELU;  W2 h + b + residual -> LayerNorm.  Pooling (masked mean or CLS)
// + L2 normalisation produce out[B,H] fp32.
#include "gemm_nt.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <mutex>
#include <vector>

// This file is compiled TWICE (kirag_amd/csrc/Makefile): once per 16-bit operand type of the MFMAs and of every stored activation —
//   bf16 (8 significand bits, fp32 range)  and  f16 (11 significand bits, |x| <= 65504; -DKR_ENC_F16)
// — into the namespaces kr::enc_bf16 / kr::enc_f16; encoder_api.hip holds the C entry points and picks one per handle (kr_encoder_create_ex).
// Why both exist: DESIGN.md section 4.2 "operand precision" (weights with outlier channels need the 11 bits to stay within the 1e-3 score tolerance).
#ifdef KR_ENC_BUILD_F16
#define KR_ENC_NS enc_f16
#else
#define KR_ENC_NS enc_bf16
#endif

namespace kr {
namespace KR_ENC_NS {

#ifdef KR_ENC_BUILD_F16
using ET = F16;
#else
using ET = BF16;
#endif

using ShapeBig = GemmShape<256, 256, 2, 4>;     // 8 waves of 128x64, 128 KiB LDS, one block per CU: best main loop (long-K GEMMs)
// Four main loops, one per launch size (launch_proj picks; all give bit-identical rows):
//   256x256 ping-pong (gemm_nt_pingpong)          launches with >= 5/8 of the CUs' worth of 256x256 tiles
//   128x128 producer / consumer (gemm_nt_split)   fewer: at most one tile per CU, or more than two
//   128x128 streaming, 2 slots, 2 blocks per CU   in between (k_proj<.., ShapeSmall, 2>)
//   32x32 / 64x64 skinny (gemm_nt_skinny)         a handful of token rows (<= 4 tiles of 32x32 per CU)

using ShapeSmall = GemmShape<128, 128, 2, 2>;   // 4 waves of 64x64, 64 KiB ring, two blocks per CU: for launches with too few 256x256 tiles to fill the chip

// all main loops run with exchanged MFMA operands (accumulators hold 4 consecutive features per lane)
template <class ShapeE, int STAGES, bool ANT = false, class Coord, class Epilogue>
__device__ __forceinline__ void gemm_main(const uint16_t* __restrict__ A, int64_t lda, int64_t M, const uint16_t* __restrict__ B, int64_t ldb, int64_t N,
                                          int K, int64_t total_tiles, char* smem, Coord&& coord, Epilogue&& epi) {
    if constexpr (ShapeE::BM == 256 && ShapeE::BN == 256) gemm_nt_pingpong<ET, true, ANT>(A, lda, M, B, ldb, N, K, total_tiles, smem, coord, epi);
    else gemm_nt_stream<ET, ShapeE, STAGES, true>(A, lda, M, B, ldb, N, K, total_tiles, smem, coord, epi);
}

struct LayerW {
    uint16_t *wqkv = nullptr, *wo = nullptr, *w1 = nullptr, *w2 = nullptr;   // bf16 [out, in]
    float *bqkv = nullptr, *bo = nullptr, *bo_eff = nullptr, *b1 = nullptr, *b2 = nullptr;   // bo_eff = bo + Wo.bv
    float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
};

// A/B switches of the projection / attention launches, read from the environment ONCE per forward (enqueue_forward), not per launch: a forward is 96
// projection launches and a small batch is bound by the host's launch rate (tests change KIRAG_AMD_PROJ_TILE between forwards of one handle, so
// reading them at kr_encoder_create would be too early)
struct Knobs {
    int pw = 8;            // KIRAG_AMD_PATCH_W: feature tiles per XCD patch of the tile walk (profiles/r03)
    int epi_prio = 0;      // KIRAG_AMD_EPI_PRIO (proj_epilogue)
    int store_nt = -1;     // KIRAG_AMD_STORE_NT = 0 / 1 forces the epilogue store policy, -1 = by output size (ProjArgs::nt)
    bool nt_h = true;      // KIRAG_AMD_NT_H=0: FF2's activation operand with plain loads
    int force_tile = 0;    // KIRAG_AMD_PROJ_TILE: 32 / 128 / 130 / 256 force a projection path (tests run every parity case through all of them)
    int ratio8 = 5;        // KIRAG_AMD_SMALL_RATIO: eighths of the CU count below which the 128x128 tiling is used (tools/ab_encoder.py)
    bool attn_lds = false, attn_dma = false;   // KIRAG_AMD_ATTN_LDS / KIRAG_AMD_ATTN_DMA: force one attention kernel (A/B)
    void read() {
        auto geti = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
        pw = geti("KIRAG_AMD_PATCH_W", 8); if (pw < 1) pw = 8;
        epi_prio = geti("KIRAG_AMD_EPI_PRIO", 0);
        store_nt = geti("KIRAG_AMD_STORE_NT", -1);
        nt_h = geti("KIRAG_AMD_NT_H", 1) != 0;
        force_tile = geti("KIRAG_AMD_PROJ_TILE", 0);
        ratio8 = geti("KIRAG_AMD_SMALL_RATIO", 5);
        attn_lds = getenv("KIRAG_AMD_ATTN_LDS") != nullptr;
        attn_dma = geti("KIRAG_AMD_ATTN_DMA", 0) != 0;
    }
};

struct Encoder {
    kr_bert_cfg cfg{};
    Knobs kn;
    int device = 0;
    float *word = nullptr, *pos = nullptr, *type = nullptr, *elng = nullptr, *elnb = nullptr;
    std::vector<LayerW> L;
    std::vector<uint8_t> got;     // 5 + 16*layers flags
    bool ready = false;
    float* stage = nullptr; size_t stage_elems = 0;   // fp32 upload staging for load_weight
    // workspace
    int64_t capT = 0; int capB = 0; int64_t capBS = 0; int64_t ldv = 0;
    int64_t *d_ids = nullptr, *d_mask = nullptr, *d_tt = nullptr;   // d_tt: token_type_ids of the batch (only written when the caller passes them)
    int *seq_off = nullptr, *seq_nk = nullptr, *seq_nq = nullptr, *seq_cls = nullptr, *seq_has0 = nullptr, *d_T = nullptr, *d_err = nullptr;
    int *tok_id = nullptr, *tok_pos = nullptr, *tok_type = nullptr;
    float *out = nullptr;
    uint8_t *xlo = nullptr;    // low half of the residual stream, one byte per element (lo_encode): written by every LayerNorm with use_lo, else by the last one only
    bool use_lo = false;       // KIRAG_AMD_RESIDUAL_LO=1 at kr_encoder_create
    uint16_t *y = nullptr, *xb = nullptr, *q = nullptr, *k = nullptr, *vT = nullptr, *ctx = nullptr, *h = nullptr;
    int lastB = 0, lastS = 0;
    int h_pad = 0;                   // extra elements per row of h (row pitch not a power of two: see ensure_ws)
    int* h_err = nullptr;            // pinned: copy of d_err taken at the end of the last asynchronous forward
    hipEvent_t ev_done = nullptr;    // recorded after that copy
    bool pending = false;            // an asynchronous forward's error word has not been looked at yet
    hipStream_t last_stream = nullptr;
    int num_cu = 256;         // CUs the persistent projection grids are sized for
    int num_cu_all = 256;     // CUs of the device (grids of the memory-bound kernels)
    struct GraphEntryT { uint64_t key; int calls; hipGraphExec_t exec; };
    std::vector<GraphEntryT> graphs;   // captured forwards of small batch shapes (run_forward)
    hipStream_t gstream = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr;
    bool graphs_off = false;
    // CLS pooling: the LAST layer's attention output / FFN only matter for one row per sequence.  Those rows are gathered into compact [B, ...] buffers
    // after the last attention and the rest of the layer runs on B rows instead of T (KIRAG_AMD_CLS_FULL=1 at kr_encoder_create: all rows, A/B and tests)
    bool cls_shortcut = true, last_shortcut = false;
    uint16_t *c_ctx = nullptr, *c_xb = nullptr, *c_y = nullptr, *c_h = nullptr; uint8_t* c_xlo = nullptr;
    int *c_off = nullptr, *c_nk = nullptr, *c_cls = nullptr, *d_B = nullptr;
};

typedef Encoder::GraphEntryT GraphEntry;
static void drop_graphs(Encoder* e) {
    for (auto& g : e->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    e->graphs.clear();
}

// ---------------------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------------------
__global__ void k_f32_to_bf16(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = ET::from_f32(src[i] * scale);
}
__global__ void k_scale_copy(const float* __restrict__ src, float* __restrict__ dst, int64_t n, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] * scale;
}

// b_o' = b_o + W_o . b_v : softmax rows sum to 1, so P (X W_v^T + b_v) = P X W_v^T + b_v and the value bias moves into the output projection
__global__ void k_fold_vbias(const uint16_t* __restrict__ wo, const float* __restrict__ bo, const float* __restrict__ bv, float* __restrict__ out, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H) return;
    float s = 0.f;
    for (int j = 0; j < H; ++j) s += ET::to_f32(wo[(int64_t)i * H + j]) * bv[j];
    out[i] = bo[i] + s;
}

// one wave per sequence: number of attended positions and whether position 0 is attended
__global__ __launch_bounds__(64) void k_seq_len(const int64_t* __restrict__ mask, int B, int S, int* __restrict__ nk, int* __restrict__ has0) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int c = 0;
    for (int p = lane; p < S; p += 64) c += (mask[(int64_t)b * S + p] != 0) ? 1 : 0;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if (lane == 0) { nk[b] = c; has0[b] = (mask[(int64_t)b * S] != 0) ? 1 : 0; }
}

// one wave: nq = nk (+1 query-only row for position 0 when CLS pooling needs it), offsets = exclusive scan of round_up(nq,4)
__global__ __launch_bounds__(64) void k_seq_scan(const int* __restrict__ nk, const int* __restrict__ has0, int B, int pool, int align, int* __restrict__ nq,
                                                 int* __restrict__ off, int* __restrict__ cls, int* __restrict__ T, int* __restrict__ err) {
    const int lane = threadIdx.x;
    int carry = 0;
    for (int base = 0; base < B; base += 64) {
        const int b = base + lane;
        int n = 0;
        if (b < B) {
            n = nk[b] + ((pool == KR_POOL_CLS && !has0[b]) ? 1 : 0);
            nq[b] = n;
            cls[b] = has0[b] ? 0 : nk[b];
        }
        const int padded = (n + align - 1) & ~(align - 1);   // align = 4, or 8 when the long-sequence attention kernel stages V^T by 16-byte LDS-DMA
        int incl = padded;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (b < B) off[b] = carry + incl - padded;
        carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) { *T = carry; }   // *err is sticky: set by k_fill_tokens, cleared by the host once it has been reported
}

// one wave per sequence: packed token list (attended positions in order; the optional query-only row for position 0 last)
// tt: token_type_ids of the batch or nullptr (= all zero, what every KiRAG caller passes); a value outside [0, type_vocab) sets error bit 2 and is read as 0
__global__ __launch_bounds__(64) void k_fill_tokens(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, const int64_t* __restrict__ tt, int S, int vocab,
                                                    int type_vocab, int align, const int* __restrict__ off, const int* __restrict__ nk, const int* __restrict__ nq,
                                                    int* __restrict__ tok_id, int* __restrict__ tok_pos, int* __restrict__ tok_type, int* __restrict__ err) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int o = off[b];
    int run = 0;
    for (int base = 0; base < S; base += 64) {
        const int p = base + lane;
        const bool v = (p < S) && (mask[(int64_t)b * S + p] != 0);
        const unsigned long long bal = __ballot(v);
        if (v) {
            const int r = run + __popcll(bal & ((1ull << lane) - 1ull));
            int64_t id = ids[(int64_t)b * S + p];
            if (id < 0 || id >= vocab) { atomicOr(err, 1); id = 0; }
            int64_t ty = tt ? tt[(int64_t)b * S + p] : 0;
            if (ty < 0 || ty >= type_vocab) { atomicOr(err, 4); ty = 0; }
            tok_id[o + r] = (int)id; tok_pos[o + r] = p; tok_type[o + r] = (int)ty;
        }
        run += __popcll(bal);
    }
    const int n = nq[b];
    if (lane == 0 && n > nk[b]) {
        int64_t id = ids[(int64_t)b * S];
        if (id < 0 || id >= vocab) { atomicOr(err, 1); id = 0; }
        int64_t ty = tt ? tt[(int64_t)b * S] : 0;
        if (ty < 0 || ty >= type_vocab) { atomicOr(err, 4); ty = 0; }
        tok_id[o + nk[b]] = (int)id; tok_pos[o + nk[b]] = 0; tok_type[o + nk[b]] = (int)ty;
    }
    const int padded = (n + align - 1) & ~(align - 1);
    if (lane < padded - n) { tok_id[o + n + lane] = 0; tok_pos[o + n + lane] = 0; tok_type[o + n + lane] = 0; }
}

// the three kernels above as ONE single-block launch for small batches (B <= PACK_SMALL_B; a 32-token forward is launch-bound, round 5): wave w counts,
// then wave 0 scans, then wave w fills — the same per-sequence code in the same order, two launches less per forward
constexpr int PACK_SMALL_B = 64;
__global__ __launch_bounds__(1024) void k_pack_small(const int64_t* __restrict__ ids, const int64_t* __restrict__ mask, const int64_t* __restrict__ tt, int B, int S, int vocab,
                                                     int type_vocab, int pool, int align, int* __restrict__ nk, int* __restrict__ has0, int* __restrict__ nq, int* __restrict__ off,
                                                     int* __restrict__ cls, int* __restrict__ T, int* __restrict__ tok_id, int* __restrict__ tok_pos, int* __restrict__ tok_type,
                                                     int* __restrict__ err) {
    __shared__ int s_nk[PACK_SMALL_B], s_h0[PACK_SMALL_B], s_nq[PACK_SMALL_B], s_off[PACK_SMALL_B];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = wave; b < B; b += 16) {                       // k_seq_len
        int c = 0;
        for (int p = lane; p < S; p += 64) c += (mask[(int64_t)b * S + p] != 0) ? 1 : 0;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
        if (lane == 0) { const int h = (mask[(int64_t)b * S] != 0) ? 1 : 0; s_nk[b] = c; s_h0[b] = h; nk[b] = c; has0[b] = h; }
    }
    __syncthreads();
    if (wave == 0) {                                           // k_seq_scan (B <= 64: one pass)
        const int b = lane;
        int n = 0;
        if (b < B) { n = s_nk[b] + ((pool == KR_POOL_CLS && !s_h0[b]) ? 1 : 0); nq[b] = n; s_nq[b] = n; cls[b] = s_h0[b] ? 0 : s_nk[b]; }
        const int padded = (n + align - 1) & ~(align - 1);
        int incl = padded;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (b < B) { off[b] = incl - padded; s_off[b] = incl - padded; }
        if (lane == 63) *T = incl;
    }
    __syncthreads();
    for (int b = wave; b < B; b += 16) {                       // k_fill_tokens
        const int o = s_off[b];
        int run = 0;
        for (int base = 0; base < S; base += 64) {
            const int p = base + lane;
            const bool v = (p < S) && (mask[(int64_t)b * S + p] != 0);
            const unsigned long long bal = __ballot(v);
            if (v) {
                const int r = run + __popcll(bal & ((1ull << lane) - 1ull));
                int64_t id = ids[(int64_t)b * S + p];
                if (id < 0 || id >= vocab) { atomicOr(err, 1); id = 0; }
                int64_t ty = tt ? tt[(int64_t)b * S + p] : 0;
                if (ty < 0 || ty >= type_vocab) { atomicOr(err, 4); ty = 0; }
                tok_id[o + r] = (int)id; tok_pos[o + r] = p; tok_type[o + r] = (int)ty;
            }
            run += __popcll(bal);
        }
        const int n = s_nq[b];
        if (lane == 0 && n > s_nk[b]) {
            int64_t id = ids[(int64_t)b * S];
            if (id < 0 || id >= vocab) { atomicOr(err, 1); id = 0; }
            int64_t ty = tt ? tt[(int64_t)b * S] : 0;
            if (ty < 0 || ty >= type_vocab) { atomicOr(err, 4); ty = 0; }
            tok_id[o + s_nk[b]] = (int)id; tok_pos[o + s_nk[b]] = 0; tok_type[o + s_nk[b]] = (int)ty;
        }
        const int padded = (n + align - 1) & ~(align - 1);
        if (lane < padded - n) { tok_id[o + n + lane] = 0; tok_pos[o + n + lane] = 0; tok_type[o + n + lane] = 0; }
    }
}

// ---- ragged input (kr_encoder_forward_packed): ids32 = the attended tokens of every sequence back to back, lens[b] = how many belong to sequence b, at
// positions 0 .. lens[b]-1 (what a right-padding tokenizer produces: collators.py:59-81 with padding=True).  These kernels fill exactly the tables the
// padded-input kernels above fill for the equivalent [B,S] batch (mask[b,p] = p < lens[b]), so everything behind them is the same code on the same data.
// A length outside [0, S], or lengths that do not add up to `total`, set error bit 8 and the sequence is read as empty (nothing is read out of bounds).
__device__ __forceinline__ void rag_fill_one(const int* __restrict__ src, int n_in, int n, int o, int vocab, int align, int lane, int* __restrict__ tok_id,
                                             int* __restrict__ tok_pos, int* __restrict__ tok_type, int* __restrict__ err) {
    for (int p = lane; p < n_in; p += 64) {
        int id = src[p];
        if (id < 0 || id >= vocab) { atomicOr(err, 1); id = 0; }
        tok_id[o + p] = id; tok_pos[o + p] = p; tok_type[o + p] = 0;
    }
    // CLS pooling of an empty sequence: the query-only row for position 0 reads token id 0 (the padded call with input_ids padded by [PAD] = 0)
    if (lane == 0 && n > n_in) { tok_id[o + n_in] = 0; tok_pos[o + n_in] = 0; tok_type[o + n_in] = 0; }
    const int padded = (n + align - 1) & ~(align - 1);
    if (lane < padded - n) { tok_id[o + n + lane] = 0; tok_pos[o + n + lane] = 0; tok_type[o + n + lane] = 0; }
}

// one wave: k_seq_len + k_seq_scan of the ragged form, plus the exclusive scan of the raw lengths (where each sequence starts in ids32)
__global__ __launch_bounds__(64) void k_rag_scan(const int* __restrict__ lens, int B, int S, int total, int pool, int align, int* __restrict__ nk, int* __restrict__ has0,
                                                 int* __restrict__ nq, int* __restrict__ off, int* __restrict__ cls, int* __restrict__ in_off, int* __restrict__ T,
                                                 int* __restrict__ err) {
    const int lane = threadIdx.x;
    int carry = 0, carry_in = 0;
    bool bad = false;
    for (int base = 0; base < B; base += 64) {
        const int b = base + lane;
        int len = (b < B) ? lens[b] : 0;
        if (len < 0 || len > S) { bad = true; len = 0; }
        int incl_in = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl_in, d, 64); if (lane >= d) incl_in += t; }
        const int start = carry_in + incl_in - len;
        if (start + len > total) { bad = true; len = 0; }          // never read past the caller's buffer (the following starts keep the caller's lengths)
        int n = 0;
        if (b < B) {
            n = len + ((pool == KR_POOL_CLS && len == 0) ? 1 : 0);
            nk[b] = len; has0[b] = len > 0 ? 1 : 0; nq[b] = n; cls[b] = 0; in_off[b] = start;
        }
        const int padded = (n + align - 1) & ~(align - 1);
        int incl = padded;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (b < B) off[b] = carry + incl - padded;
        carry += __shfl(incl, 63, 64);
        carry_in += __shfl(incl_in, 63, 64);
    }
    if (__ballot(bad) != 0ull || carry_in != total) { if (lane == 0) atomicOr(err, 8); }
    if (lane == 0) *T = carry;
}

__global__ __launch_bounds__(64) void k_rag_fill(const int* __restrict__ ids32, int vocab, int align, const int* __restrict__ in_off, const int* __restrict__ off,
                                                 const int* __restrict__ nk, const int* __restrict__ nq, int* __restrict__ tok_id, int* __restrict__ tok_pos,
                                                 int* __restrict__ tok_type, int* __restrict__ err) {
    const int b = blockIdx.x;
    rag_fill_one(ids32 + in_off[b], nk[b], nq[b], off[b], vocab, align, threadIdx.x, tok_id, tok_pos, tok_type, err);
}

// both as ONE single-block launch for B <= PACK_SMALL_B (the ragged twin of k_pack_small)
__global__ __launch_bounds__(1024) void k_rag_small(const int* __restrict__ ids32, const int* __restrict__ lens, int B, int S, int total, int vocab, int pool, int align,
                                                    int* __restrict__ nk, int* __restrict__ has0, int* __restrict__ nq, int* __restrict__ off, int* __restrict__ cls,
                                                    int* __restrict__ T, int* __restrict__ tok_id, int* __restrict__ tok_pos, int* __restrict__ tok_type, int* __restrict__ err) {
    __shared__ int s_nk[PACK_SMALL_B], s_nq[PACK_SMALL_B], s_off[PACK_SMALL_B], s_in[PACK_SMALL_B];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        const int b = lane;
        int len = (b < B) ? lens[b] : 0;
        bool bad = false;
        if (len < 0 || len > S) { bad = true; len = 0; }
        int incl_in = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl_in, d, 64); if (lane >= d) incl_in += t; }
        const int start = incl_in - len;
        if (start + len > total) { bad = true; len = 0; }
        int n = 0;
        if (b < B) {
            n = len + ((pool == KR_POOL_CLS && len == 0) ? 1 : 0);
            nk[b] = len; has0[b] = len > 0 ? 1 : 0; nq[b] = n; cls[b] = 0; s_nk[b] = len; s_nq[b] = n; s_in[b] = start;
        }
        const int padded = (n + align - 1) & ~(align - 1);
        int incl = padded;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (b < B) { off[b] = incl - padded; s_off[b] = incl - padded; }
        const int sum_in = __shfl(incl_in, 63, 64);
        if (__ballot(bad) != 0ull || sum_in != total) { if (lane == 0) atomicOr(err, 8); }
        if (lane == 63) *T = incl;
    }
    __syncthreads();
    for (int b = wave; b < B; b += 16) rag_fill_one(ids32 + s_in[b], s_nk[b], s_nq[b], s_off[b], vocab, align, lane, tok_id, tok_pos, tok_type, err);
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
// two floats -> packed 16-bit pair (lo in bits 0..15), round-to-nearest-even, NaN stays NaN: ONE v_cvt_pk_bf16_f32 / the f16 conversions of the target
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#ifdef KR_ENC_BUILD_F16
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, f16x2_t));
#else
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2_t));
#endif
}
// the two halves of a packed pair as fp32
__device__ __forceinline__ float unpack_lo16(unsigned int w) {
#ifdef KR_ENC_BUILD_F16
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu));
#else
    return __builtin_bit_cast(float, w << 16);
#endif
}
__device__ __forceinline__ float unpack_hi16(unsigned int w) {
#ifdef KR_ENC_BUILD_F16
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
#else
    return __builtin_bit_cast(float, w & 0xffff0000u);
#endif
}

// Low half of the residual stream: x = hi + lo with hi = the 16-bit operand the GEMMs read (round-to-nearest of x) and lo = ONE byte: the distance from hi to
// x in units of ulp(hi) / 256, biased by 128 (19 significand bits with f16 operands, 16 with bf16).  Round 2 kept lo as a second 16-bit word (10 B per
// element through a LayerNorm instead of 6); the byte makes it 8 B — on the emulation of the rounding points (tools/precision_probe.py, G10 out3 / out16) the
// worst score error is unchanged (9.1e-5 vs 8.6e-5 / 1.2e-3 vs 9.5e-4).
// The byte is computed ON THE BIT PATTERNS (round 4): x and hi have the same sign and neighbouring magnitudes, so bits(x) - bits(hi) (both as fp32) is their
// distance in fp32 ulps, signed towards larger magnitude, and 2^LO_SH of those are ulp(hi) / 256 (a unit half as large when hi rounded up into the next
// binade: still within +-128).  Encode = subtract, add the rounding constant, shift, clamp (4 integer instructions); decode = shift-add + constant (3, the
// byte extraction included) — a third of the floating-point form (exponent extraction, two constructed powers of two, rint, min, max, conversions) that made
// round 4's fused residual epilogue VALU-bound (profiles/r04/tried_fused_layernorm.txt).  |x| < 2^-25 (hi = +-0) clamps to a denormal: an absolute error below 2^-25.
#ifdef KR_ENC_BUILD_F16
constexpr int LO_SH = 5;        // 23 - 10 stored mantissa bits - 8
#else
constexpr int LO_SH = 8;        // 23 - 7 - 8
#endif
__device__ __forceinline__ unsigned int lo_encode(float o, float hf) {
    const int d = (int)(__builtin_bit_cast(unsigned int, o) - __builtin_bit_cast(unsigned int, hf));
    // round to nearest, biased by 128; clamped BEFORE the shift.  (Shift-then-clamp of two neighbours is selected to v_ashr_pk_u8_i32, which on gfx950 writes only
    // bits 15:0 of its destination — the compiler assumes the upper half is zero and ORs the stale bits into the neighbouring bytes: tools/ashr_pk_check.hip)
    const int t = min(max(d + ((1 << (LO_SH - 1)) + (128 << LO_SH)), 0), (256 << LO_SH) - 1);
    return (unsigned int)t >> LO_SH;
}
__device__ __forceinline__ float lo_decode(unsigned int byte, float hf) {
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned int, hf) + (byte << LO_SH) - (128u << LO_SH));
}
// lo_decode for readers of the FINAL hidden state (pooling, kr_encoder_last_hidden): a non-finite hi (a LayerNorm output beyond the f16 range, or NaN) decodes
// to ITSELF — on the bit patterns inf - 4096 would be a finite 3.4e38.  Inside the stack the overflow travels with the 16-bit stream itself (the next GEMM reads
// hi = inf and every LayerNorm behind it sees NaN through y), so the LayerNorm's residual decode keeps the 3-instruction form (its two extra VALU instructions per
// element cost 0.3 % of the 1000-query step); behind the LAST LayerNorm there is no GEMM, only this decode (tests/test_lo_codec_spec.py,
// tests/test_gpu_lifecycle.py: an overflow in the last LayerNorm must raise KR_ERANGE).
__device__ __forceinline__ float lo_decode_final(unsigned int byte, float hf) {
    const unsigned int hb = __builtin_bit_cast(unsigned int, hf);
    return (hb & 0x7f800000u) == 0x7f800000u ? hf : lo_decode(byte, hf);
}

// LayerNorm of one row held as up to 8 float4 per lane (H <= 2048); writes fp32 and bf16 copies
__device__ __forceinline__ void ln_row_store(float4 (&v)[8], int H, int lane, const float* __restrict__ g, const float* __restrict__ bta, float eps,
                                             uint8_t* __restrict__ xlo_row, uint16_t* __restrict__ xb_row) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (lane * 4 + j * 256 < H) s += v[j].x + v[j].y + v[j].z + v[j].w;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    const float mu = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (lane * 4 + j * 256 < H) {
            const float a = v[j].x - mu, b = v[j].y - mu, c = v[j].z - mu, d = v[j].w - mu;
            q += a * a + b * b + c * c + d * d;
        }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
    const float rstd = 1.0f / sqrtf(q / (float)H + eps);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = lane * 4 + j * 256;
        if (i < H) {
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            const float4 bb = *reinterpret_cast<const float4*>(bta + i);
            float4 o;
            o.x = (v[j].x - mu) * rstd * gg.x + bb.x; o.y = (v[j].y - mu) * rstd * gg.y + bb.y;
            o.z = (v[j].z - mu) * rstd * gg.z + bb.z; o.w = (v[j].w - mu) * rstd * gg.w + bb.w;
            ushort4 ob;
            ob.x = ET::from_f32(o.x); ob.y = ET::from_f32(o.y); ob.z = ET::from_f32(o.z); ob.w = ET::from_f32(o.w);
            *reinterpret_cast<ushort4*>(xb_row + i) = ob;
            if (xlo_row)                                                      // optional low half (kernel-uniform branch): one byte per element
                *reinterpret_cast<unsigned int*>(xlo_row + i) = lo_encode(o.x, ET::to_f32(ob.x)) | (lo_encode(o.y, ET::to_f32(ob.y)) << 8) |
                                                                (lo_encode(o.z, ET::to_f32(ob.z)) << 16) | (lo_encode(o.w, ET::to_f32(ob.w)) << 24);
        }
    }
}

// embeddings: word[id] + position[pos] + token_type[0] -> LayerNorm       (one wave per token)
__global__ __launch_bounds__(256) void k_embed_ln(const int* __restrict__ tok_id, const int* __restrict__ tok_pos, const int* __restrict__ tok_type, const int* __restrict__ Tp,
                                                  const float* __restrict__ word, const float* __restrict__ pos, const float* __restrict__ type,
                                                  const float* __restrict__ g, const float* __restrict__ bta, float eps, int H,
                                                  uint8_t* __restrict__ xlo, uint16_t* __restrict__ xb) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= *Tp) return;
    const float* w = word + (int64_t)tok_id[t] * H;
    const float* p = pos + (int64_t)tok_pos[t] * H;
    const float* ty = type + (int64_t)tok_type[t] * H;       // token_type_embeddings row (0 for every KiRAG caller)
    float4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = lane * 4 + j * 256;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < H) {
            const float4 a = *reinterpret_cast<const float4*>(w + i);
            const float4 b = *reinterpret_cast<const float4*>(p + i);
            const float4 c = *reinterpret_cast<const float4*>(ty + i);
            v[j] = make_float4((a.x + b.x) + c.x, (a.y + b.y) + c.y, (a.z + b.z) + c.z, (a.w + b.w) + c.w);
        }
    }
    ln_row_store(v, H, lane, g, bta, eps, xlo ? xlo + t * H : nullptr, xb + t * H);
}

// LayerNorm(y + bias + residual) with 16-byte accesses: a lane owns 8 consecutive elements per 512-element step (one global_load_dwordx4 per tensor and step:
// 8-byte accesses reach 0.54-0.70 of the 16-byte rate, MI355X_MICROARCH.md).  NS 512-element steps cover a row (H <= 512 NS, H % 8 == 0).
// The row arithmetic lives in Ln16<NS> (one definition for every kernel that normalises a row).
template <int NS>
struct Ln16 {
    float gg[NS][8], bb[NS][8], yb[NS][8];     // gamma, beta, the projection's bias (added in fp32) of the lane's elements
    __device__ __forceinline__ void load_params(const float* __restrict__ g, const float* __restrict__ bta, const float* __restrict__ ybias, int H, int lane) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int i = lane * 8 + j * 512;
#pragma unroll
            for (int c = 0; c < 8; ++c) { gg[j][c] = 0.f; bb[j][c] = 0.f; yb[j][c] = 0.f; }
            if (i < H) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 a = *reinterpret_cast<const float4*>(g + i + 4 * h), b = *reinterpret_cast<const float4*>(bta + i + 4 * h), c = *reinterpret_cast<const float4*>(ybias + i + 4 * h);
                    gg[j][4 * h] = a.x; gg[j][4 * h + 1] = a.y; gg[j][4 * h + 2] = a.z; gg[j][4 * h + 3] = a.w;
                    bb[j][4 * h] = b.x; bb[j][4 * h + 1] = b.y; bb[j][4 * h + 2] = b.z; bb[j][4 * h + 3] = b.w;
                    yb[j][4 * h] = c.x; yb[j][4 * h + 1] = c.y; yb[j][4 * h + 2] = c.z; yb[j][4 * h + 3] = c.w;
                }
            }
        }
    }
    // v = (y + bias) + residual, the residual decoded from its 16-bit half and its low-half byte
    __device__ __forceinline__ void combine(const uint4 (&a)[NS], const uint4 (&rh)[NS], const uint2 (&rl)[NS], float (&v)[NS][8]) const {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const unsigned int aw[4] = {a[j].x, a[j].y, a[j].z, a[j].w}, hw[4] = {rh[j].x, rh[j].y, rh[j].z, rh[j].w}, lw[2] = {rl[j].x, rl[j].y};
#pragma unroll
            for (int c = 0; c < 4; ++c) {   // elements 2c, 2c + 1: low-half bytes 2c, 2c + 1 of the 8
                v[j][2 * c] = (unpack_lo16(aw[c]) + yb[j][2 * c]) + lo_decode((lw[c >> 1] >> (16 * (c & 1))) & 0xffu, unpack_lo16(hw[c]));
                v[j][2 * c + 1] = (unpack_hi16(aw[c]) + yb[j][2 * c + 1]) + lo_decode((lw[c >> 1] >> (16 * (c & 1) + 8)) & 0xffu, unpack_hi16(hw[c]));
            }
        }
    }
    // mean / variance over the wave, normalise, store the 16-bit row and (xlo_row != nullptr) its low-half bytes.  NTS: the low half stored non-temporally
    template <bool NTS>
    __device__ __forceinline__ void normalize_store(const float (&v)[NS][8], int H, float eps, int lane, uint16_t* xb_row, uint8_t* xlo_row) const {
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (lane * 8 + j * 512 < H)
#pragma unroll
                for (int c = 0; c < 8; ++c) s += v[j][c];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        const float mu = s / (float)H;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NS; ++j)
            if (lane * 8 + j * 512 < H)
#pragma unroll
                for (int c = 0; c < 8; ++c) { const float dlt = v[j][c] - mu; q += dlt * dlt; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) q += __shfl_xor(q, m, 64);
        const float rstd = 1.0f / sqrtf(q / (float)H + eps);
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int i = lane * 8 + j * 512;
            if (i < H) {
                float o[8];
                unsigned int ob[4], ol[4];
#pragma unroll
                for (int c = 0; c < 8; ++c) o[c] = (v[j][c] - mu) * rstd * gg[j][c] + bb[j][c];
#pragma unroll
                for (int c = 0; c < 4; ++c) ob[c] = pack_bf16x2(o[2 * c], o[2 * c + 1]);
                *reinterpret_cast<uint4*>(xb_row + i) = make_uint4(ob[0], ob[1], ob[2], ob[3]);
                if (xlo_row) {   // optional low half: one byte per element (lo_encode)
                    ol[0] = ol[1] = 0u;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        ol[c >> 1] |= (lo_encode(o[2 * c], unpack_lo16(ob[c])) | (lo_encode(o[2 * c + 1], unpack_hi16(ob[c])) << 8)) << (16 * (c & 1));
                    if constexpr (NTS) __builtin_nontemporal_store(u32x2_t{ol[0], ol[1]}, reinterpret_cast<u32x2_t*>(xlo_row + i));
                    else *reinterpret_cast<uint2*>(xlo_row + i) = make_uint2(ol[0], ol[1]);
                }
            }
        }
    }
};

// one wave per token row, grid-stride.  Same arithmetic per element as round 1's 8-byte k_ln; the row sums add the elements in the lane order of Ln16
// (rows stay independent of the batch).
// POL (cache policy of the streams; profiles/r04/tried_ln_policies.txt): bit 0 = y loaded non-temporally (dead after this kernel), bit 1 = the low half loaded
// non-temporally, bit 2 = the low half stored non-temporally (its next reader is the next LayerNorm, ~600 MiB of traffic later)
template <int NS, int POL = 0>
__global__ __launch_bounds__(256) void k_ln16(const uint16_t* __restrict__ y, const float* __restrict__ ybias, const int* __restrict__ Tp, const float* __restrict__ g,
                                              const float* __restrict__ bta, float eps, int H, const uint8_t* xlo_in, uint8_t* xlo, uint16_t* xb) {
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
    const int lane = threadIdx.x & 63;
    const int T = *Tp;
    Ln16<NS> ln;
    ln.load_params(g, bta, ybias, H, lane);
    const int64_t step = (int64_t)gridDim.x * 4;
    int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint4 a[NS], rh[NS]; uint2 rl[NS];
    auto load_row = [&](int64_t row) {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int i = lane * 8 + j * 512;
            a[j] = rh[j] = make_uint4(0u, 0u, 0u, 0u); rl[j] = make_uint2(0x80808080u, 0x80808080u);      // byte 128 = a zero low half
            if (i < H && row < T) {
                a[j] = (POL & 1) ? __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(y + row * H + i))) : *reinterpret_cast<const uint4*>(y + row * H + i);
                rh[j] = *reinterpret_cast<const uint4*>(xb + row * H + i);
                if (xlo_in) rl[j] = (POL & 2) ? __builtin_bit_cast(uint2, __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(xlo_in + row * H + i))) : *reinterpret_cast<const uint2*>(xlo_in + row * H + i);
            }
        }
    };
    load_row(t);
    for (; t < T; t += step) {
        float v[NS][8];
        ln.combine(a, rh, rl, v);
        load_row(t + step);                                   // next row's loads in flight while this one is reduced and stored
        ln.template normalize_store<(POL & 4) != 0>(v, H, eps, lane, xb + t * H, xlo ? xlo + t * H : nullptr);
    }
}

// ---------------------------------------------------------------------------------------------------------
// projections: C[token, feature] = X[token, :] . W[feature, :]   (rows = tokens, cols = output features)
// ---------------------------------------------------------------------------------------------------------
struct ProjArgs {
    const uint16_t* W; const uint16_t* X; const int* Tp; int F; int K; int H;
    const float* bias;
    uint16_t* out0; uint16_t* out1; uint16_t* outT; int64_t ldT;   // QKV: q, k row-major [T,H]; vT [H, ldT].  Others: out0 [T, F]
    int64_t ldx, ldo;   // row pitch (elements) of X and of out0 (EPI_DENSE / EPI_GELU); 0 = K / F
    int pw;   // feature tiles per XCD patch of the tile walk (patch_coord)
    int epi_prio;   // A/B knob, see proj_epilogue
    int ant;  // activation loads non-temporal (FF2's h: launch_proj)
    int nt;   // epilogue stores non-temporal (large launches: the output is consumed from HBM by the next kernel, keep it out of L2) or plain
              // (small launches: the whole output fits in L2 / Infinity Cache, the next kernel reads it from there)
};

enum { EPI_QKV = 0, EPI_DENSE = 1, EPI_GELU = 2 };

// erf-GELU x Phi(x) = max(x, 0) - 0.5 |x| erfc(|x| / sqrt 2), with erfc(a / sqrt 2) = 2^-Q(a), Q(a) = a (c1 + c2 a + c3 a^2 + c4 a^3 + c5 a^4) a weighted
// minimax fit of -log2 erfc on [0, 8] (weight a erfc(a / sqrt 2) = the sensitivity of the result; fitted offline, c5 > 0 so Q keeps growing and the
// tail underflows to the exact limit max(x, 0)).  |error| <= 9.4e-7 absolute on the whole line in fp32 arithmetic (the result is rounded to bf16:
// 2^-9 relative), no sign handling, and ONE quarter-rate transcendental (v_exp_f32) per element instead of two: 13 VALU instructions per element
// pair (76 issue cycles) against 22 (136) for the Abramowitz-Stegun 7.1.26 form used before, in an epilogue that nothing overlaps with (the GELU
// was 1557 VALU instructions per wave and 256x256 tile, with both waves of a SIMD in it at the same time).
// two elements at once: the polynomial / products run as packed fp32 (v_pk_fma_f32 / v_pk_mul_f32), only exp2 / abs / max stay scalar
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) {
    const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
    f32x2 q = __builtin_elementwise_fma(ax, f32x2{4.881049150e-04f, 4.881049150e-04f}, f32x2{-7.198719129e-03f, -7.198719129e-03f});
    q = __builtin_elementwise_fma(q, ax, f32x2{5.214659068e-02f, 5.214659068e-02f});
    q = __builtin_elementwise_fma(q, ax, f32x2{4.595959239e-01f, 4.595959239e-01f});
    q = __builtin_elementwise_fma(q, ax, f32x2{1.151000505e+00f, 1.151000505e+00f});
    q = q * ax;
    const f32x2 e = {__builtin_amdgcn_exp2f(-q.x), __builtin_amdgcn_exp2f(-q.y)};
    const f32x2 r = {fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
    return __builtin_elementwise_fma(ax * e, f32x2{-0.5f, -0.5f}, r);
}

// A 32x32 MFMA accumulator has its COLUMN on the lane, so a direct store writes 2-byte elements (128 store instructions per lane per
// 256x256 tile, the epilogue then costs as much as a third of the main loop).  Instead every wave owns a private 4-KiB LDS stage
// behind the ring (no barrier: only this wave touches it, LDS operations of one wave complete in order):
//   rows:  the wave's 32 x 64 block (one mi, both ni) is written as bf16 [32 rows][128 B] and read back 16 B per lane, so each global
//          store instruction writes eight whole 128-B lines of the output;
//   V^T:   each 32x32 tile is written TRANSPOSED ([feature][token], 4 consecutive tokens of a lane packed into 8 B, 80-B rows) and
//          read back 16 B per lane: a store instruction writes 64-B runs of sixteen V^T rows.
constexpr int EPI_STAGE_BYTES = 4096;

// Both helpers take the SWAPPED accumulator layout of gemm_nt_pingpong / gemm_nt_split / gemm_nt_stream / gemm_nt_skinny with SWAP = true: tile (mi, ni), register r,
// lane (c = l & 31, h = l >> 5) is token mi*32 + c, feature ni*32 + (r & 3) + 8 (r >> 2) + 4 h of the wave's (TM*32 tokens) x (TN*32 features).
//
// rows: for one mi the wave's 32 tokens x 64 features are staged as bf16 [32 tokens][128 B]; registers 4g .. 4g+3 of a lane are 4
// consecutive features -> one packed ds_write_b64 (16-B chunk index XOR (token & 7): 2-way instead of 16-way conflicts), read back 16 B
// per lane: every global store instruction writes eight whole 128-B rows.  f(v, mi, ni, g) maps 4 features (bias / GELU / row scale) before packing.
template <class Shape, bool NT, class F>
__device__ __forceinline__ void store_rows_bf16(AccTile<Shape>& acc, char* stage, uint16_t* __restrict__ out, int64_t ld, int64_t row0, int col0, F&& f) {
    static_assert(Shape::TN == 2, "stage geometry assumes 64 features per wave");
    const int c = acc.lane & 31, h = acc.lane >> 5;
    const int r8 = acc.lane >> 3, ch = acc.lane & 7;
    const char* st_rd = stage + r8 * 128 + ((ch ^ r8) << 4);
    uint16_t* g_base = out + (row0 + r8) * ld + col0 + ch * 8;
    // software pipeline over the mi blocks: write(mi), read(mi), THEN the global stores of mi-1 — LDS operations of one wave complete in order, so
    // the single 4-KiB stage is safe to overwrite right after the reads were issued, and the stores of block mi-1 only wait for their own reads
    // (counted lgkmcnt) while the LDS round trip of block mi is in flight (one exposed round trip per tile instead of one per block)
    uint4 d[2][4];
#pragma unroll
    for (int mi = 0; mi <= Shape::TM; ++mi) {
#ifdef KR_STAMP
        acc.stamp(mi);                      // [0] = everything before the first block (bias loads ...), [mi] = block mi-1
#endif
        if (mi < Shape::TM) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = f(f32x4{acc.v[mi][ni][4 * g], acc.v[mi][ni][4 * g + 1], acc.v[mi][ni][4 * g + 2], acc.v[mi][ni][4 * g + 3]}, mi, ni, g);
                    uint2 w;
                    w.x = pack_bf16x2(v.x, v.y); w.y = pack_bf16x2(v.z, v.w);
                    *reinterpret_cast<uint2*>(stage + c * 128 + (((ni * 4 + g) ^ (c & 7)) << 4) + h * 8) = w;
                }
#pragma unroll
            for (int p = 0; p < 4; ++p) d[mi & 1][p] = *reinterpret_cast<const uint4*>(st_rd + p * 8 * 128);
        }
        if (mi > 0) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {  // rl & 7 == lane >> 3 for every p: one lane-dependent LDS / global base, the rest are wave-uniform steps (8 rows per store)
                u32x4_t* dst = reinterpret_cast<u32x4_t*>(g_base + (int64_t)((mi - 1) * 32 + p * 8) * ld);
                if constexpr (NT) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, d[(mi - 1) & 1][p]), dst);
                else *dst = __builtin_bit_cast(u32x4_t, d[(mi - 1) & 1][p]);
            }
        }
    }
#ifdef KR_STAMP
    acc.stamp(Shape::TM + 1);
#endif
}

// V^T[feature, token]: each 32x32 tile is staged as [32 features][32 tokens] (80-B rows), lanes = consecutive tokens of a feature row,
// read back 16 B per lane: a store instruction writes 64-B runs of sixteen V^T rows.
template <class Shape, bool NT>
__device__ __forceinline__ void store_transposed_bf16(AccTile<Shape>& acc, char* stage, uint16_t* __restrict__ outT, int64_t ldT, int64_t t0, int f0) {
    // the lane id is made opaque HERE: everything below that depends on it (LDS offsets, the 64-bit V^T addresses) is then recomputed per tile (a few VALU
    // instructions) instead of being hoisted out of the persistent tile loop into registers the main loop has no room for — hipcc spilled them, and the
    // scratch reloads (VMEM, followed by s_waitcnt vmcnt(0)) drained the LDS-DMA ring in every V^T tile (tests/test_capi_and_host.py: no spills allowed)
    int ln = acc.lane;
    asm volatile("" : "+v"(ln));
    const int c = ln & 31, h = ln >> 5;
#pragma unroll
    for (int mi = 0; mi < Shape::TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < Shape::TN; ++ni) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                *reinterpret_cast<uint16_t*>(stage + ((r & 3) + 8 * (r >> 2) + 4 * h) * 80 + c * 2) = ET::from_f32(acc.v[mi][ni][r]);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int fl = p * 16 + (ln >> 2), ch = ln & 3;
                const uint4 d = *reinterpret_cast<const uint4*>(stage + fl * 80 + ch * 16);
                u32x4_t* dst = reinterpret_cast<u32x4_t*>(outT + (int64_t)(f0 + ni * 32 + fl) * ldT + t0 + mi * 32 + ch * 8);
                if constexpr (NT) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, d), dst);
                else *dst = __builtin_bit_cast(u32x4_t, d);
            }
        }
}

// persistent streaming projections (256x256 ping-pong or 128x128 streaming main loop).  rows = tokens, cols = output features; bias is
// one value per lane and ni.  Token-indexed buffers are allocated in multiples of 256 rows, so a partial last token tile needs no bounds
// test (rows >= T are written with values computed from clamped loads and never read).
// Tiles are walked in patches of (token tiles x 8 feature tiles) per XCD so operand slices are reused from that XCD's L2.
//   EPI_QKV:   F = 3H: features [0,H) -> q (bias, 1/8 folded into the weights), [H,2H) -> k, [2H,3H) -> V^T (its bias lives in bo_eff)
//   EPI_DENSE: out0[T,F] = acc as bf16 (k_ln adds the bias and the residual in fp32)
//   EPI_GELU:  out0[T,F] = gelu(acc + bias)
template <int EPI, class ShapeE, bool NT>
__device__ __forceinline__ void proj_epilogue(const ProjArgs& a, AccTile<ShapeE>& acc, int64_t m0, int64_t n0, char* stage) {
    const int64_t t0 = m0 + acc.m_wave;
    const int f0 = (int)n0 + acc.n_wave;          // first feature of this wave's 64 columns; F % 64 == 0, so a wave is never partial
    if (f0 >= a.F) return;
    // A/B knob (KIRAG_AMD_EPI_PRIO, profiles/r03/tried_ab_epi_prio.txt): the two wave groups of the ping-pong loop run their epilogues side by side and the
    // younger group (tile rows 128 ..) loses the issue arbitration (its epilogue takes ~2x as long): 1 = that group at priority 1, 2 = the older group
    if (a.epi_prio && ((a.epi_prio == 1) == (acc.m_wave >= 128))) __builtin_amdgcn_s_setprio(1);
    const int h = acc.lane >> 5;
    f32x4 b[2][4];                                // bias of the lane's 32 features: (ni, g) -> features ni*32 + 8g + 4h .. +3
    if constexpr (EPI != EPI_DENSE) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) b[ni][g] = *reinterpret_cast<const f32x4*>(a.bias + f0 + ni * 32 + 8 * g + 4 * h);
    }
    if constexpr (EPI == EPI_QKV) {
        const int region = f0 / a.H;              // H % 64 == 0: a wave's columns never straddle q | k | v
        if (region == 2) {
            store_transposed_bf16<ShapeE, NT>(acc, stage, a.outT, a.ldT, t0, f0 - 2 * a.H);   // value bias lives in bo_eff
        } else {
            store_rows_bf16<ShapeE, NT>(acc, stage, region ? a.out1 : a.out0, a.H, t0, f0 - region * a.H,
                                    [&](f32x4 v, int mi, int ni, int g) { return v + b[ni][g]; });
        }
    } else if constexpr (EPI == EPI_DENSE) {
        store_rows_bf16<ShapeE, NT>(acc, stage, a.out0, a.ldo, t0, f0, [&](f32x4 v, int, int, int) { return v; });   // the bias is added in k_ln (fp32)
    } else {
        store_rows_bf16<ShapeE, NT>(acc, stage, a.out0, a.ldo, t0, f0, [&](f32x4 v, int mi, int ni, int g) {
            const f32x4 x = v + b[ni][g];
            const f32x2 lo = gelu_erf_fast2(f32x2{x.x, x.y}), hi = gelu_erf_fast2(f32x2{x.z, x.w});
            return f32x4{lo.x, lo.y, hi.x, hi.y};
        });
    }
    if (a.epi_prio) __builtin_amdgcn_s_setprio(0);
}

// ANT: the activation operand is loaded non-temporally (FF2's h: launch_proj)
template <int EPI, class ShapeE, int STAGES, bool NT, bool ANT = false>
__global__ __launch_bounds__(ShapeE::NTHREADS, 2) void k_proj(ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = *a.Tp;
    const int64_t tm_count = (T + ShapeE::BM - 1) / ShapeE::BM, tn_count = (a.F + ShapeE::BN - 1) / ShapeE::BN;
    char* stage = smem + STAGES * ShapeE::STAGE_BYTES + (threadIdx.x >> 6) * EPI_STAGE_BYTES;
    auto coord = [&](int64_t nat, int64_t& m0, int64_t& n0) {
        int64_t tm, tn;
        patch_coord(nat, tm_count, tn_count, tm, tn, (uint32_t)a.pw);
        m0 = tm * ShapeE::BM; n0 = tn * ShapeE::BN;
    };
    gemm_main<ShapeE, STAGES, ANT>(a.X, a.ldx, T, a.W, a.K, a.F, a.K, tm_count * tn_count, smem, coord,
                                   [&](AccTile<ShapeE>& acc, int64_t m0, int64_t n0, int64_t) { proj_epilogue<EPI, ShapeE, NT>(a, acc, m0, n0, stage); });
}

// the same projections on the producer / consumer 128x128 loop (gemm_nt_split): 4 multiplying + 4 staging waves, 4-slot ring + one 4-KiB epilogue
// stage per multiplying wave = 144 KiB, one persistent block per CU
template <int EPI>
__global__ __launch_bounds__(SPLIT_THREADS) void k_proj_split(ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = *a.Tp;
    const int64_t tm_count = (T + 127) / 128, tn_count = (a.F + 127) / 128;
    char* stage = smem + SPLIT_RING * ShapeSplit::STAGE_BYTES + ((threadIdx.x >> 6) & 3) * EPI_STAGE_BYTES;
    gemm_nt_split<ET, true>(
        a.X, a.ldx, T, a.W, a.K, a.F, a.K, tm_count * tn_count, smem,
        [&](int64_t nat, int64_t& m0, int64_t& n0) {
            int64_t tm, tn;
            patch_coord(nat, tm_count, tn_count, tm, tn, (uint32_t)a.pw);
            m0 = tm * 128; n0 = tn * 128;
        },
        [&](AccTile<ShapeSplit>& acc, int64_t m0, int64_t n0, int64_t) { proj_epilogue<EPI, ShapeSplit, false>(a, acc, m0, n0, stage); });
}

// the same projections for a handful of token rows on the skinny loop (gemm_nt_skinny): one (32 WM)-token x (32 WN)-feature tile per block,
// grid = (F / (32 WN), T / (32 WM)).  Epilogue straight from the accumulator (swapped layout: a lane holds 4 consecutive features of one token per register
// quad): 8-byte row stores, 2-byte stores for V^T — at these sizes the stores are noise next to the operand stream.
// (Round 5 built the LayerNorm behind a dense projection as the TAIL of this launch — write-through y, an arrival counter per token tile, the last
// arriver normalises the tile's rows — bit-identical and SLOWER: the one block that finds itself last works through 32 rows alone, 26.8 us per launch
// against 9.2 + 5.3 for the two launches; profiles/r05/tried_ln_tail.txt.)
template <int EPI, int RING, int WM = 1, int WN = 1>
__global__ __launch_bounds__((WM * WN + 4) * 64) void k_proj_skinny(ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int64_t m0 = (int64_t)blockIdx.y * (32 * WM), n0 = (int64_t)blockIdx.x * (32 * WN);
    f32x4 bias4[4];                                       // multiplying waves: the bias of the lane's 16 features, requested before the main loop
    gemm_nt_skinny<ET, RING, WM, WN, true>(a.X, a.ldx, a.Tp, m0, a.W, a.K, a.F, n0, a.K, smem,
        [&](int64_t, int64_t f0) {
            if constexpr (EPI != EPI_DENSE) {
                const int h = (threadIdx.x & 63) >> 5;
#pragma unroll
                for (int g = 0; g < 4; ++g) bias4[g] = *reinterpret_cast<const f32x4*>(a.bias + f0 + 8 * g + 4 * h);   // QKV: the V third's slots are never used
            }
        },
        [&](AccTile<ShapeSkinny>& acc, int64_t t0, int64_t f0) {
        const int c = acc.lane & 31, h = acc.lane >> 5;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc.v[0][0][4 * g], acc.v[0][0][4 * g + 1], acc.v[0][0][4 * g + 2], acc.v[0][0][4 * g + 3]};
            const int f = (int)f0 + 8 * g + 4 * h;        // first of the lane's 4 consecutive features
            if constexpr (EPI == EPI_QKV) {
                const int region = (int)f0 / a.H;
                if (region == 2) {                        // V^T [feature, token]; its bias lives in bo_eff
#pragma unroll
                    for (int j = 0; j < 4; ++j) a.outT[(int64_t)(f - 2 * a.H + j) * a.ldT + t0 + c] = ET::from_f32(v[j]);
                    continue;
                }
                v = v + bias4[g];
                uint2 w; w.x = pack_bf16x2(v.x, v.y); w.y = pack_bf16x2(v.z, v.w);
                *reinterpret_cast<uint2*>((region ? a.out1 : a.out0) + (t0 + c) * a.H + (f - region * a.H)) = w;
            } else {
                if constexpr (EPI == EPI_GELU) {
                    v = v + bias4[g];
                    const f32x2 lo = gelu_erf_fast2(f32x2{v.x, v.y}), hi = gelu_erf_fast2(f32x2{v.z, v.w});
                    v = f32x4{lo.x, lo.y, hi.x, hi.y};
                }
                uint2 w; w.x = pack_bf16x2(v.x, v.y); w.y = pack_bf16x2(v.z, v.w);
                *reinterpret_cast<uint2*>(a.out0 + (t0 + c) * a.ldo + f) = w;
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------------------
// attention.  Two kernels share ONE arithmetic (attn_step64: the same MFMAs, the same softmax operations in the same order), so a sequence's context
// rows are bit-identical whichever kernel its batch selects (the embedding cache and batch invariance rely on it):
//   k_attn_lds  sequences of at most 128 tokens: K / V^T of the block's heads staged through registers, one block per (sequence, HPB heads, 4 / HPB q-tiles)
//   k_attn_dma  longer sequences: K / V^T chunks of 64 keys stream through a 3-stage LDS ring by LDS-DMA (two chunks in flight behind the one being
//               multiplied, one barrier per chunk), one block per (sequence, head, 8 q-tiles): two q-tiles per wave share every staged chunk
// Common scheme per wave and 32-query tile: S^T = K.Q^T (keys on accumulator rows, so the softmax row reductions are in-lane + one shfl_xor 32), online
// softmax over 64-key steps, P^T fed from the accumulator straight into the V^T.P^T MFMA.  The key that sits on A-tile row i of a 32-key tile is
// perm(i) = i with bits 2 and 3 exchanged: a lane's registers 8a .. 8a+7 then hold 8 CONSECUTIVE keys (16 a + 8 hf .. + 7), i.e. the P^T fragment of a
// k-step matches one contiguous 16-byte run of a V^T row (without the permutation a lane owns keys {0..3, 8..11} + 4 hf: two 8-byte reads per fragment).
// The permutation maps each ds_read_b128 lane group onto itself, so the K reads stay bank-conflict free.
//   K image   128-B rows, 16-B chunk index XOR ((key >> 1) & 7) (same image as the GEMM ring)
//   V^T image k_attn_lds: row pitch 2 * cap + 8 bytes (pitch / 8 odd: conflict-free ds_read_b64), keys >= nk stored as zero;
//             k_attn_dma: [64 d][128 B] per chunk, chunk index XOR ((d >> 1) & 7) like the K image (one ds_read_b128 per fragment), columns >= nk of
//             the last chunk zeroed in LDS after they landed (no 0 * NaN from rows of other sequences).
// The O tile is staged through a wave-private 4-KiB LDS block and stored as whole 128-B rows of ctx.
// ---------------------------------------------------------------------------------------------------------
struct AttnState {
    f32x16 o0, o1;
    float mref, l;      // reference maximum (log2 units) the accumulated o / l are scaled by; running denominator
};

__device__ __forceinline__ void attn_init(AttnState& s) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { s.o0[r] = 0.f; s.o1[r] = 0.f; }
    s.mref = -INFINITY; s.l = 0.f;
}

__device__ __forceinline__ int attn_perm(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

// One 64-key step of a wave's 32-query tile.  Kst: K image (128-B rows, swizzled), k0 = row of the step's first key in it; nvalid = keys of the
// step that exist (MASKED instantiation: < 64; <= 32: the second 32-key tile is skipped — it would only add exact zeros); vfrag(dh, ks) returns the
// V^T A-fragment of d rows 32 dh .. + 31, keys k0 + 16 ks + 8 hf .. + 7.
// Scores are in log2 units (log2(e) / sqrt(d_h) is folded into W_q at load time), so P = exp2(s - mref) is one subtraction and one v_exp_f32 per
// element.  The reference maximum mref of a query is only raised — with the o / l rescale of the online softmax — when a step's maximum exceeds it by
// more than ATTN_RESCALE_THR (2^8: P stays far inside the 16-bit range, and o / l are fp32); any mref gives the same result up to rounding because it
// cancels in o / l.  The slow path is taken by the whole wave (wave-uniform branch), so a tile's P.V is never split.  (Feeding -mref to the S^T MFMAs
// as their C operand would save the subtraction too, but costs 16 more live registers per tile: measured as spills at three blocks per CU.)
constexpr float ATTN_RESCALE_THR = 8.0f;
// first half of a step: the shifted-score tiles S^T = K . Q^T (st1 = -inf when the second 32-key tile does not exist)
// MODE (wave-uniform, picked from the number of valid keys of the step): 0 = 64, 1 = fewer than 32 (first tile masked, no second tile),
// 2 = exactly 32 (one full tile: a 32-token sequence), 3 = 33 .. 63 (second tile masked)
// PF (k_attn_dma): all K fragments of the step are requested before the first MFMA (counted lgkmcnt waits instead of one exposed LDS round trip per
// MFMA); same MFMAs in the same order, so the result does not depend on it
template <int MODE, bool PF = false>
__device__ __forceinline__ void attn_scores(f32x16& st0, f32x16& st1, const uint4 (&qf)[4], const char* Kst, int k0, int nvalid, int c, int hf) {
    constexpr bool two = MODE == 0 || MODE == 3;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (PF) {
        const int key = k0 + attn_perm(c);
        const char* krow = Kst + key * 128;
        const int swz = (key >> 1) & 7;
        uint4 kf[two ? 8 : 4];
#if defined(ATTN_ABL) && ATTN_ABL == 7      // ablation: no LDS fragment reads (fragments = registers)
#pragma unroll
        for (int sk = 0; sk < (two ? 8 : 4); ++sk) kf[sk] = qf[sk & 3];
        (void)krow; (void)swz;
#else
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) kf[sk] = *reinterpret_cast<const uint4*>(krow + (((2 * sk + hf) ^ swz) << 4));
        if constexpr (two) {
#pragma unroll
            for (int sk = 0; sk < 4; ++sk) kf[4 + sk] = *reinterpret_cast<const uint4*>(krow + 32 * 128 + (((2 * sk + hf) ^ swz) << 4));
        }
#endif
        st0 = ET::mfma(kf[0], qf[0], zero);
#pragma unroll
        for (int sk = 1; sk < 4; ++sk) st0 = ET::mfma(kf[sk], qf[sk], st0);
        if constexpr (two) {
            st1 = ET::mfma(kf[4], qf[0], zero);
#pragma unroll
            for (int sk = 1; sk < 4; ++sk) st1 = ET::mfma(kf[4 + sk], qf[sk], st1);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) st1[r] = -INFINITY;
        }
    } else {
        const int key = k0 + attn_perm(c);
        const char* krow = Kst + key * 128;
        const int swz = (key >> 1) & 7;
        st0 = ET::mfma(*reinterpret_cast<const uint4*>(krow + (((0 + hf) ^ swz) << 4)), qf[0], zero);
#pragma unroll
        for (int sk = 1; sk < 4; ++sk) st0 = ET::mfma(*reinterpret_cast<const uint4*>(krow + (((2 * sk + hf) ^ swz) << 4)), qf[sk], st0);
        if (two) {
            const char* krow1 = krow + 32 * 128;          // (key + 32) >> 1 & 7 == swz
            st1 = ET::mfma(*reinterpret_cast<const uint4*>(krow1 + (((0 + hf) ^ swz) << 4)), qf[0], zero);
#pragma unroll
            for (int sk = 1; sk < 4; ++sk) st1 = ET::mfma(*reinterpret_cast<const uint4*>(krow1 + (((2 * sk + hf) ^ swz) << 4)), qf[sk], st1);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) st1[r] = -INFINITY;      // never read (two == false below)
        }
    }
    // register r of this lane: key k0 + 16 (r >> 3) + 8 hf + (r & 7) (+ 32 for st1), query c.  Only the tile that holds key nvalid is partial (wave-uniform
    // cases: a 32-token sequence has exactly one full tile and nothing to mask)
    if constexpr (MODE == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st0[r] = (16 * (r >> 3) + 8 * hf + (r & 7) < nvalid) ? st0[r] : -INFINITY;
    }
    if constexpr (MODE == 3) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st1[r] = (16 * (r >> 3) + 8 * hf + (r & 7) + 32 < nvalid) ? st1[r] : -INFINITY;
    }
}

// second half: online softmax and O^T += V^T . P^T
template <int MODE, bool PF = false, class VFrag>
__device__ __forceinline__ void attn_softmax_pv(AttnState& s, const f32x16& st0, const f32x16& st1, VFrag&& vfrag) {
    constexpr bool two = MODE == 0 || MODE == 3;
    uint4 vpre[PF ? (two ? 8 : 4) : 1];
    if constexpr (PF) {                                  // the V^T fragments land under the maximum / rescale arithmetic
#pragma unroll
#if defined(ATTN_ABL) && ATTN_ABL == 7
        for (int ks = 0; ks < (two ? 4 : 2); ++ks) { vpre[2 * ks] = make_uint4(__builtin_bit_cast(unsigned, st0[ks]), 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); vpre[2 * ks + 1] = vpre[2 * ks]; }
#else
        for (int ks = 0; ks < (two ? 4 : 2); ++ks) { vpre[2 * ks] = vfrag(0, ks); vpre[2 * ks + 1] = vfrag(1, ks); }
#endif
    }
#if defined(ATTN_ABL) && (ATTN_ABL == 2 || ATTN_ABL == 6 || ATTN_ABL == 7)     // tools/attn_bench.hip ablation: no maximum
    float tmax = st0[0];
#else
    float tmax = fmaxf(fmaxf(st0[0], st0[1]), st0[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) tmax = fmaxf(fmaxf(tmax, st0[r]), st0[r + 1]);   // v_max3_f32
    tmax = fmaxf(tmax, st0[15]);
    if (two) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) tmax = fmaxf(fmaxf(tmax, st1[r]), st1[r + 1]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
#endif
    const bool fresh = s.mref == -INFINITY;               // nothing accumulated for this query yet
    if (__builtin_amdgcn_ballot_w64(fresh || tmax > s.mref + ATTN_RESCALE_THR) != 0ull) {
        // raise the reference (never lower it)
        const float mnew = fmaxf(s.mref, tmax);
        const float alpha = fresh ? 1.f : __builtin_amdgcn_exp2f(s.mref - mnew);   // fresh: o = l = 0 (and mref - mnew is -inf - x)
        s.l *= alpha;
        s.o0 *= alpha; s.o1 *= alpha;
        s.mref = mnew;
    }
    const float mshift = (s.mref == -INFINITY) ? 0.f : s.mref;   // still -inf: no valid key so far, every score is -inf and stays so
    // P = exp2(s - mref) and O^T += V^T . P^T, one k-step (8 keys per lane: registers 8 a .. 8 a + 7 = keys 16 a + 8 hf .. + 7 of the tile) at a time: the
    // exp2 / pack of a k-step sit between the MFMAs of the previous one, and at most 8 probabilities are live next to the scores
    float psum = 0.f;
    auto pv = [&](const f32x16& stx, int ks0) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float e[8];
#pragma unroll
#if defined(ATTN_ABL) && (ATTN_ABL == 6 || ATTN_ABL == 7)     // ablation: no softmax arithmetic at all (P = S)
            for (int i = 0; i < 8; ++i) { e[i] = stx[8 * a + i]; }
            psum = e[0];
#elif defined(ATTN_ABL) && ATTN_ABL == 1     // ablation: no exponential
            for (int i = 0; i < 8; ++i) { e[i] = stx[8 * a + i] - mshift; psum += e[i]; }
#elif defined(ATTN_ABL) && ATTN_ABL == 3   // ablation: no row sum
            for (int i = 0; i < 8; ++i) { e[i] = __builtin_amdgcn_exp2f(stx[8 * a + i] - mshift); }
            psum = e[0];
#else
            for (int i = 0; i < 8; ++i) { e[i] = __builtin_amdgcn_exp2f(stx[8 * a + i] - mshift); psum += e[i]; }
#endif
            uint4 pf;
            pf.x = pack_bf16x2(e[0], e[1]); pf.y = pack_bf16x2(e[2], e[3]); pf.z = pack_bf16x2(e[4], e[5]); pf.w = pack_bf16x2(e[6], e[7]);
#if defined(ATTN_ABL) && ATTN_ABL == 4     // ablation: no P.V MFMAs (P and the V^T fragments stay live)
            if constexpr (PF) {
                const uint4 va = vpre[2 * (ks0 + a)], vb = vpre[2 * (ks0 + a) + 1];
                const unsigned keep = pf.x ^ pf.y ^ pf.z ^ pf.w ^ va.x ^ va.y ^ va.z ^ va.w ^ vb.x ^ vb.y ^ vb.z ^ vb.w;
                asm volatile("" :: "v"(keep));
            } else
#endif
            if constexpr (PF) {
                s.o0 = ET::mfma(vpre[2 * (ks0 + a)], pf, s.o0);
                s.o1 = ET::mfma(vpre[2 * (ks0 + a) + 1], pf, s.o1);
            } else {
                s.o0 = ET::mfma(vfrag(0, ks0 + a), pf, s.o0);
                s.o1 = ET::mfma(vfrag(1, ks0 + a), pf, s.o1);
            }
        }
    };
    pv(st0, 0);
    if (two) pv(st1, 2);
#if !(defined(ATTN_ABL) && (ATTN_ABL == 3 || ATTN_ABL == 6 || ATTN_ABL == 7))
    psum += __shfl_xor(psum, 32, 64);
#endif
    s.l += psum;
}

template <int MODE, bool PF = false, class VFrag>
__device__ __forceinline__ void attn_step64m(AttnState& s, const uint4 (&qf)[4], const char* Kst, int k0, int nvalid, int c, int hf, VFrag&& vfrag) {
    f32x16 st0, st1;
    attn_scores<MODE, PF>(st0, st1, qf, Kst, k0, nvalid, c, hf);
    attn_softmax_pv<MODE, PF>(s, st0, st1, vfrag);
}
// MASKED = false: 64 valid keys; true: fewer (nvalid says how many)
template <bool MASKED, bool PF = false, class VFrag>
__device__ __forceinline__ void attn_step64(AttnState& s, const uint4 (&qf)[4], const char* Kst, int k0, int nvalid, int c, int hf, VFrag&& vfrag) {
    if constexpr (!MASKED) attn_step64m<0, PF>(s, qf, Kst, k0, 64, c, hf, vfrag);
    else if (nvalid == 32) attn_step64m<2>(s, qf, Kst, k0, nvalid, c, hf, vfrag);
    else if (nvalid < 32) attn_step64m<1>(s, qf, Kst, k0, nvalid, c, hf, vfrag);
    else attn_step64m<3>(s, qf, Kst, k0, nvalid, c, hf, vfrag);
}

// normalise a finished 32-query tile and store it as whole 128-B rows of ctx through the wave-private 4-KiB LDS block Os
__device__ __forceinline__ void attn_store_tile(const AttnState& s, char* Os, uint16_t* __restrict__ ctx, int64_t off, int q0, int nq, int H, int head, int lane) {
    asm volatile("" : "+v"(lane));   // opaque: the store addresses are computed here, after the key loop, instead of living in registers (or scratch) across it
    const int c = lane & 31, hf = lane >> 5;
    // a query with no attendable key (all-masked sequence) is 0/0 = NaN, as under HF's -inf masking
    const float inv = 1.0f / s.l;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        uint2 w0, w1;
        w0.x = pack_bf16x2(s.o0[4 * gq + 0] * inv, s.o0[4 * gq + 1] * inv); w0.y = pack_bf16x2(s.o0[4 * gq + 2] * inv, s.o0[4 * gq + 3] * inv);
        w1.x = pack_bf16x2(s.o1[4 * gq + 0] * inv, s.o1[4 * gq + 1] * inv); w1.y = pack_bf16x2(s.o1[4 * gq + 2] * inv, s.o1[4 * gq + 3] * inv);
        const int j8 = 2 * gq + hf;     // 8-byte chunk (4 features) of the 128-B row of query c
        *reinterpret_cast<uint2*>(Os + c * 128 + ((j8 ^ (c & 15)) << 3)) = w0;
        *reinterpret_cast<uint2*>(Os + c * 128 + (((8 + j8) ^ (c & 15)) << 3)) = w1;
    }
#pragma unroll
    for (int p4 = 0; p4 < 4; ++p4) {
        const int rq = p4 * 8 + (lane >> 3), ch = lane & 7;
        const uint2 lo = *reinterpret_cast<const uint2*>(Os + rq * 128 + (((2 * ch) ^ (rq & 15)) << 3));
        const uint2 hi = *reinterpret_cast<const uint2*>(Os + rq * 128 + (((2 * ch + 1) ^ (rq & 15)) << 3));
        if (q0 + rq < nq) *reinterpret_cast<uint4*>(ctx + (off + q0 + rq) * H + head * 64 + ch * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

#ifndef ALDS_PF_N
#define ALDS_PF_N 0
#endif
constexpr bool ALDS_PF = ALDS_PF_N != 0;                   // fragment prefetch in k_attn_lds (tools: A/B builds)
// HPB = heads per block: 1 when a sequence has >= 3 q-tiles, 2 / 4 for short sequences so that all four waves have work.
// __launch_bounds__(256, 3): at most 168 registers per lane, which makes hipcc keep the MFMA accumulators in VGPRs; with the default bound it put
// them in AGPRs and spent 112 of the 276 VALU instructions of a key tile on v_accvgpr_read / _write around the softmax rescale
template <int HPB, bool NTL = false>   // NTL (experiment builds only): q / k / v loaded non-temporally
__global__ __launch_bounds__(256, 3) void k_attn_lds(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ vT, int64_t ldv,
                                                  const int* __restrict__ seq_off, const int* __restrict__ seq_nk, const int* __restrict__ seq_nq,
                                                  int H, int heads, int kchunk, uint16_t* __restrict__ ctx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction; as a scalar, the head / q-tile / LDS bases derived from it cost no VGPRs
    const int hb = blockIdx.x, b = blockIdx.y;            // heads fastest: the heads of one sequence (same 2-KiB q/k rows) run together
    const int nq = seq_nq[b];
    constexpr int QT = 4 / HPB;                           // q-tiles per block (one per wave and head)
    const int q0 = ((int)blockIdx.z * QT + wave / HPB) * 32;
    if ((int)blockIdx.z * QT * 32 >= nq) return;          // block-uniform: no q-tile of this block exists (nq == 0 included)
    const int nk = seq_nk[b];
    const int64_t off = seq_off[b];
    const int vpitch = kchunk * 2 + 8;
    char* Ks = smem;                                      // [HPB][kchunk][128 B]
    char* Vs = smem + (size_t)HPB * kchunk * 128;         // [HPB][64][vpitch]
    char* Os = Vs + (size_t)HPB * 64 * vpitch + wave * 4096;
    const int hs = wave % HPB;
    const int head = hb * HPB + hs;
    const bool active = head < heads && q0 < nq;          // inactive waves still stage and meet every barrier
    const char* Kh = Ks + (size_t)hs * kchunk * 128;
    const char* Vh = Vs + (size_t)hs * 64 * vpitch;
    const int c = lane & 31, hf = lane >> 5;
    // Q^T as the B operand: lane (c, hf) holds Q[q0 + c][16 s + 8 hf .. +7], s = 0..3
    uint4 qf[4] = {};
    if (active) {
        const int qi = (q0 + c < nq) ? (q0 + c) : (nq - 1);
        const uint16_t* qrow = q + (off + qi) * H + head * 64;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = NTL ? __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(qrow + 16 * s + 8 * hf)))
                                                : *reinterpret_cast<const uint4*>(qrow + 16 * s + 8 * hf);
    }
    AttnState st;
    attn_init(st);
    constexpr int NB = HPB == 1 ? 4 : 2;                  // loads in flight per thread, head and batch (register budget: 3 blocks per CU = 168 VGPRs; HPB = 4 means <= 32 keys: 2 cover a head)
    for (int kc0 = 0; kc0 < nk; kc0 += kchunk) {
        // ---- stage keys [kc0, kc0 + kchunk) of the block's heads: every global load of a batch is issued before the first LDS store
        // (a load -> store loop would serialise one memory round trip per iteration)
        const int nkc = min(nk - kc0, kchunk);            // keys of this chunk
        const int nkp = (nkc + 31) & ~31;
        const int cpr = nkp >> 2;                         // 8-byte chunks (4 keys) per V^T row
        const unsigned cpr_magic = 0xFFFFFFFFu / (unsigned)cpr + 1u;
        if (kc0 > 0) __syncthreads();                     // every wave is done with the previous chunk
        {
            // all heads of the block in ONE batch: every global load (K and V^T of up to HPB heads) is issued before the first LDS store, so a block with
            // 2 / 4 heads pays one memory round trip per batch, not one per head (a 32-token sequence is a single batch)
            const int nkcs = nkp * 8, nvc = 64 * cpr;
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));      // opaque: the staging indices are derived per chunk and die with it (the key loop below runs at the register limit)
            for (int base = 0; base < nkcs || base < nvc; base += 256 * NB) {
                uint4 kv[HPB][NB]; uint2 vv[HPB][NB];
#pragma unroll
                for (int h2 = 0; h2 < HPB; ++h2) {
                    const int head2 = hb * HPB + h2;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int i = base + j * 256 + tid;
                        const int key = i >> 3, ch = i & 7;
                        kv[h2][j] = make_uint4(0u, 0u, 0u, 0u);
                        if (head2 < heads && i < nkcs && key < nkc) {
                            const uint16_t* kp = k + (off + kc0 + key) * H + head2 * 64 + ch * 8;
                            kv[h2][j] = NTL ? __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(kp))) : *reinterpret_cast<const uint4*>(kp);
                        }
                    }
                }
                // (d row, 8-byte chunk) of element i of the V^T staging: i / cpr and i % cpr (exact: i < 2^16, cpr <= 128); recomputed where needed (two VALU
                // instructions) instead of being kept in registers across the loads — the kernel runs at the 168-register limit of three blocks per CU
                auto vsplit = [&](int i, int& d, int& kc) { d = (int)__umulhi((unsigned)i, cpr_magic); kc = i - d * cpr; };
#pragma unroll
                for (int h2 = 0; h2 < HPB; ++h2) {
                    const int head2 = hb * HPB + h2;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int i = base + j * 256 + tid;
                        int vd, vk; vsplit(i, vd, vk);
                        vv[h2][j] = make_uint2(0u, 0u);
                        if (head2 < heads && i < nvc && vk * 4 < nkc) {
                            const uint16_t* vp = vT + (int64_t)(head2 * 64 + vd) * ldv + off + kc0 + vk * 4;   // off, kc0 % 4 == 0: 8-B aligned
                            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
                            vv[h2][j] = NTL ? __builtin_bit_cast(uint2, __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(vp))) : *reinterpret_cast<const uint2*>(vp);
                        }
                    }
                }
#pragma unroll
                for (int h2 = 0; h2 < HPB; ++h2) {
                    if (hb * HPB + h2 >= heads) continue;
                    char* Kw = Ks + (size_t)h2 * kchunk * 128;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int i = base + j * 256 + tid;
                        const int key = i >> 3, ch = i & 7;
                        if (i < nkcs) *reinterpret_cast<uint4*>(Kw + key * 128 + ((ch ^ ((key >> 1) & 7)) << 4)) = kv[h2][j];
                    }
                }
#pragma unroll
                for (int h2 = 0; h2 < HPB; ++h2) {
                    if (hb * HPB + h2 >= heads) continue;
                    char* Vw = Vs + (size_t)h2 * 64 * vpitch;
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const int i = base + j * 256 + tid;
                        int vd, vk; vsplit(i, vd, vk);
                        const int key0 = vk * 4;
                        uint2 v = vv[h2][j];
                        if (key0 + 4 > nkc) {      // keys >= nk (padding / the next sequence) are stored as zero
                            v.x &= (key0 + 0 < nkc ? 0xffffu : 0u) | (key0 + 1 < nkc ? 0xffff0000u : 0u);
                            v.y &= (key0 + 2 < nkc ? 0xffffu : 0u) | (key0 + 3 < nkc ? 0xffff0000u : 0u);
                        }
                        if (i < nvc) *reinterpret_cast<uint2*>(Vw + vd * vpitch + vk * 8) = v;
                    }
                }
            }
        }
        __syncthreads();
        if (!active) continue;
        // the full 64-key steps run in a loop of their own: with the masked variants inside the same loop the accumulators went through register
        // copies at every join (k0: key offset inside the chunk)
        const int kfull = nkc & ~63;
        for (int k0 = 0; k0 < kfull; k0 += 64) {
            auto vfrag = [&](int dh, int ks) {
                const char* v = Vh + (c + 32 * dh) * vpitch + (k0 + 16 * ks + 8 * hf) * 2;
                const uint2 a0 = *reinterpret_cast<const uint2*>(v), a1 = *reinterpret_cast<const uint2*>(v + 8);
                return make_uint4(a0.x, a0.y, a1.x, a1.y);
            };
            attn_step64<false, ALDS_PF>(st, qf, Kh, k0, 64, c, hf, vfrag);
        }
        if (kfull < nkc) {
            auto vfrag = [&](int dh, int ks) {
                const char* v = Vh + (c + 32 * dh) * vpitch + (kfull + 16 * ks + 8 * hf) * 2;
                const uint2 a0 = *reinterpret_cast<const uint2*>(v), a1 = *reinterpret_cast<const uint2*>(v + 8);
                return make_uint4(a0.x, a0.y, a1.x, a1.y);
            };
            attn_step64<true>(st, qf, Kh, kfull, nkc - kfull, c, hf, vfrag);
        }
    }
    if (!active) return;
    attn_store_tile(st, Os, ctx, off, q0, nq, H, head, lane);
}

// ---- long sequences: LDS-DMA ring ------------------------------------------------------------------------------------------------------------
#ifdef KR_STAMP_ATTN
__device__ unsigned long long kr_attn_stamps[8];          // diagnostic build only (tools/attn_bench.hip)
#endif
constexpr int ADMA_STAGE = 16384;                          // per 64-key chunk: K [64 keys][128 B] + V^T [64 d][128 B]
constexpr int ADMA_RING = 3;
constexpr int ADMA_LDS = ADMA_RING * ADMA_STAGE;           // 48 KiB; the O staging (4 waves x 4 KiB) re-uses the ring after the last chunk
#ifndef ADMA_WAVES_N
#define ADMA_WAVES_N 4                                     // tools/attn_bench.hip builds the 8-wave variant too (one block per (sequence, head) up to 512 tokens:
#endif                                                     // 5 % faster at 128 x 512, 10-35 % slower at 64 x 512, 256 x 256 and 341 x 192)
constexpr int ADMA_WAVES = ADMA_WAVES_N;
constexpr int ADMA_THREADS = ADMA_WAVES * 64;
constexpr int ADMA_QT = 2 * ADMA_WAVES;                    // q-tiles per block: TWO per wave (w and w + ADMA_WAVES), so every staged chunk serves 256 queries
constexpr int ADMA_PIECES = 8 / ADMA_WAVES;                // K pieces (and V^T pieces) of 1 KiB a wave issues per chunk

// Measured at 128 x 512 tokens (us per layer; the register-staged kernel: 344): two q-tiles per wave one after the other 253; one q-tile per wave with
// four waves per block and three waves per SIMD 285 (every chunk then serves 128 queries and the block meets a barrier per step: its waves run in
// lockstep); eight waves x one q-tile at <= 128 registers spills the Q fragments (scratch reloads are VMEM operations: they drain the DMA ring).
// Interleaving a wave's two q-tiles by halves (scores(0), scores(1), softmax + P.V(0), softmax + P.V(1): tile 1's S^T MFMAs under tile 0's softmax)
// needs both score tiles live: 42 spilled registers at the 256-register limit, two of them reloaded per chunk (VMEM: the DMA ring drains) — not kept.
// (the body is a function with __restrict__ K / V^T pointers on purpose: after inlining the LDS-DMA carries their alias scope and the ring's ds_reads are
// marked as not aliasing it, which lets the compiler's waitcnt pass leave the COUNTED vmcnt waits alone; see coarse_q32_body in search.hip)
__device__ __forceinline__ void attn_dma_body(const uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ vT, int64_t ldv,
                                              int64_t off, int nk, int nq, int H, int head, int qg, int64_t capT, uint16_t* __restrict__ ctx, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, hf = lane >> 5;
    const int nchunks = (nk + 63) >> 6;
    // DMA of one chunk: 16 pieces of 1 KiB (8 K pieces = 8 keys x 128 B each, 8 V^T pieces = 8 d rows x 128 B each); wave w issues pieces
    // w * ADMA_PIECES .. + ADMA_PIECES - 1 of both.  lane -> row 8 p + (lane >> 3), 16-B chunk (lane & 7) ^ swizzle(row) of that row (the LDS destination is
    // lane-linear: the swizzle sits on the source)
    // addresses = a buffer resource per operand whose base is this (sequence, head)'s first byte (scalar registers) + a wave-uniform chunk offset (the
    // instruction's scalar offset) + a 32-bit per-lane offset that never changes: buffer_load_dwordx4 ... lds.  (With global_load_lds the compiler kept four
    // zero-extended 64-bit lane offsets, spilled them at the 256-register limit and reloaded them — s_waitcnt vmcnt(0) each — in front of the DMA of the
    // partial last chunk.)  Rows past the sequence (last chunk) are read and masked; the K buffer has 64 rows and V^T 64 columns of slack behind the last
    // token (ensure_ws), so nothing is out of range.
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(k)) + (int64_t)head * 128 + off * H * 2, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(vT)) + ((int64_t)head * 64) * ldv * 2 + off * 2, 0, 0xffffffff, 0x00020000);
    uint32_t klane[ADMA_PIECES], vlane[ADMA_PIECES];
#pragma unroll
    for (int pp = 0; pp < ADMA_PIECES; ++pp) {
        const int row = 8 * (wave * ADMA_PIECES + pp) + (lane >> 3);   // key inside the chunk / d row
        const int sw = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
        klane[pp] = (uint32_t)(row * H * 2 + sw);
        vlane[pp] = (uint32_t)((int64_t)row * ldv * 2 + sw);          // < 2^32: 64 rows x (tokens + 64) x 2 B, tokens <= 2^24 (ensure_ws)
    }
    auto issue = [&](int cidx) {
        char* stg = smem + (cidx % ADMA_RING) * ADMA_STAGE;
        const int kc = cidx * 64 * H * 2, vc = cidx * 128;             // chunk offsets (bytes): <= 512 tokens per sequence
#pragma unroll
        for (int pp = 0; pp < ADMA_PIECES; ++pp) {
            const int p = wave * ADMA_PIECES + pp;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)(stg + p * 1024), 16, klane[pp], kc, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(stg + 8192 + p * 1024), 16, vlane[pp], vc, 0, 0);
        }
    };
    issue(0);
    if (nchunks > 1) issue(1);
    // The Q fragments are loaded behind the first two chunks' DMA and waited for HERE with a wait the compiler sees (a builtin, not inline asm):
    // otherwise its waitcnt pass keeps them "possibly pending" around the loop's back edge and puts s_waitcnt vmcnt(0) in front of the first MFMA of
    // every chunk, which drains the DMA ring (one memory round trip per chunk, as without a ring)
    int q0[2]; bool act[2];
    uint4 qf[2][4] = {};
    AttnState st[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        q0[j] = (qg * ADMA_QT + wave + ADMA_WAVES * j) * 32;
        act[j] = q0[j] < nq;                               // wave-uniform; act[1] implies act[0]
        if (act[j]) {
            const int qi = (q0[j] + c < nq) ? (q0[j] + c) : (nq - 1);
            const uint16_t* qrow = q + (off + qi) * H + head * 64;
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[j][s] = *reinterpret_cast<const uint4*>(qrow + 16 * s + 8 * hf);
        }
        attn_init(st[j]);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): Q fragments (and chunks 0, 1) have landed
#ifdef KR_STAMP_ATTN
    unsigned long long sa_turn = 0, sa_t0 = 0, sa_t1 = 0, sa_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long sa_begin = sa_prev;
#define KR_SA(acc) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - sa_prev; sa_prev = now_; } while (0)
#else
#define KR_SA(acc) do { } while (0)
#endif
    auto turn = [&](int ci) {
        if (ci + 1 < nchunks) wait_vmcnt<2 * ADMA_PIECES>(); else wait_vmcnt<0>();   // this wave's pieces of chunk ci have landed; chunk ci + 1 may be in flight
        __builtin_amdgcn_s_barrier();                      // everybody's pieces of chunk ci landed, everybody is done reading chunk ci - 1
#if defined(ATTN_ABL) && ATTN_ABL == 5      // tools/attn_bench.hip ablation: no K / V^T traffic after the first two chunks (stale stages are multiplied)
        (void)ci;
#else
        if (ci + 2 < nchunks) issue(ci + 2);               // into the stage of chunk ci - 1
#endif
    };
    // The chunks with 64 valid keys run in loops of their own (one per number of active q-tiles) that hold nothing but the unmasked step: with the masked
    // variants and the act[] tests inside one loop the accumulators of both tiles went through copies at every join (32 v_mov_b64 per step)
    const int nfull = nk >> 6;
    auto full_chunk = [&](int ci, auto two_tiles) {
        turn(ci);
        KR_SA(sa_turn);
        char* stg = smem + (ci % ADMA_RING) * ADMA_STAGE;
        auto vfrag = [&](int dh, int ks) {
            const int d = c + 32 * dh;
            return *reinterpret_cast<const uint4*>(stg + 8192 + d * 128 + (((2 * ks + hf) ^ ((d >> 1) & 7)) << 4));
        };
        attn_step64<false, true>(st[0], qf[0], stg, 0, 64, c, hf, vfrag);
        KR_SA(sa_t0);
        if constexpr (decltype(two_tiles)::value) attn_step64<false, true>(st[1], qf[1], stg, 0, 64, c, hf, vfrag);
        KR_SA(sa_t1);
    };
    if (act[1]) {
        for (int ci = 0; ci < nfull; ++ci) full_chunk(ci, std::true_type{});
    } else {
        for (int ci = 0; ci < nfull; ++ci) full_chunk(ci, std::false_type{});
    }
    for (int ci = nfull; ci < nchunks; ++ci) {             // at most one: the partial last chunk
        turn(ci);
        char* stg = smem + (ci % ADMA_RING) * ADMA_STAGE;
        const int nkc = min(nk - ci * 64, 64);
        auto vfrag = [&](int dh, int ks) {
            const int d = c + 32 * dh;
            return *reinterpret_cast<const uint4*>(stg + 8192 + d * 128 + (((2 * ks + hf) ^ ((d >> 1) & 7)) << 4));
        };
        if (nkc < 64) {
            // V^T columns >= nkc of the last chunk hold other sequences' values (or padding): zero them, so that P = 0 meets 0 and not a possible NaN / Inf
            for (int i = tid; i < 64 * 8; i += ADMA_THREADS) {
                const int d = i >> 3, chk = i & 7;         // (d row, 16-B chunk)
                if (chk * 8 + 8 > nkc) {
                    uint4* w = reinterpret_cast<uint4*>(stg + 8192 + d * 128 + ((chk ^ ((d >> 1) & 7)) << 4));
                    uint4 v = *w;
                    unsigned int* u = reinterpret_cast<unsigned int*>(&v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int key = chk * 8 + 2 * e;
                        u[e] &= (key < nkc ? 0xffffu : 0u) | (key + 1 < nkc ? 0xffff0000u : 0u);
                    }
                    *w = v;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (!act[j]) continue;
            if (nkc < 64) attn_step64<true>(st[j], qf[j], stg, 0, nkc, c, hf, vfrag);
            else attn_step64<false>(st[j], qf[j], stg, 0, 64, c, hf, vfrag);
        }
    }
    __syncthreads();                                       // the ring is free: re-use it for the O staging
    char* Os = smem + wave * 4096;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (act[j]) attn_store_tile(st[j], Os, ctx, off, q0[j], nq, H, head, lane);
#ifdef KR_STAMP_ATTN
    if (lane == 0) {       // [0] barrier / wait / DMA issue, [1] first tile's step, [2] second tile's step, [3] whole lifetime, [4] waves
        const unsigned long long end = __builtin_amdgcn_s_memtime();
        atomicAdd(&kr_attn_stamps[0], sa_turn); atomicAdd(&kr_attn_stamps[1], sa_t0); atomicAdd(&kr_attn_stamps[2], sa_t1);
        atomicAdd(&kr_attn_stamps[3], end - sa_begin); atomicAdd(&kr_attn_stamps[4], 1ull);
    }
#endif
}

__global__ __launch_bounds__(ADMA_THREADS, 512 / ADMA_THREADS) void k_attn_dma(const uint16_t* q, const uint16_t* k, const uint16_t* vT, int64_t ldv, const int* __restrict__ seq_off,
                                                             const int* __restrict__ seq_nk, const int* __restrict__ seq_nq, int H, int64_t capT, uint16_t* ctx, int heads, int nseq, int qgroups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // linear block id -> (pair p = b * heads + head, q-group qg).  Consecutive workgroups go to consecutive XCDs (id mod 8), and each XCD has its own L2:
    // the q-groups of one (sequence, head) read the same K / V^T stream, so they are placed 8 ids apart — same XCD, dispatched together — and the second
    // reader finds the chunks in L2 (measured: HBM-side traffic of the kernel 1.6 x -> 1.1 x its algorithmic bytes at 128 x 512 tokens).  Heads fastest
    // inside a group of 8 pairs: the heads of one sequence share their 2-KiB q / k rows.
    const int G = qgroups;                                 // q-groups per pair
    const int L = (int)blockIdx.x;
    const int grp = L / (8 * G), r = L % (8 * G);
    const int qg = r >> 3, p = grp * 8 + (r & 7);
    const int b = p / heads, head = p % heads;
    if (b >= nseq) return;                                 // the grid is padded to whole groups of 8 pairs
    const int nq = seq_nq[b];
    if (qg * ADMA_QT * 32 >= nq) return;                   // block-uniform (nq == 0 included)
    attn_dma_body(q, k, vT, ldv, seq_off[b], seq_nk[b], nq, H, head, qg, capT, ctx, smem);
}

// pooling + L2 normalisation: one block per sequence.  Mean pooling: wave w sums the tokens t = w, w+4, ... (8-byte loads of the (hi, lo)
// stream, 4 columns per lane and step), the four partial sums are combined in a fixed order (w = 0..3), so the result is deterministic.
// pooling + L2 normalisation: one block of 16 waves per sequence.  Masked mean: wave w sums tokens w, w + 16, ... (in that order), the 16 partial rows are added
// in wave order — a fixed order per sequence, whatever the batch.  (Rounds 1-4 used 4 waves: the kernel is bound by the VALU work of decoding and adding
// 262 k elements for a 256-token sequence on ONE block, 37 us — 2.7 % of a one-sequence forward; 16 waves: see profiles/r05.)
constexpr int POOL_WAVES = 16;
template <int NJ>     // 256-element steps that cover a row: H <= 256 NJ
__global__ __launch_bounds__(POOL_WAVES * 64) void k_pool(const uint16_t* __restrict__ xb, const uint8_t* __restrict__ xlo, const int* __restrict__ seq_off,
                                                         const int* __restrict__ seq_nk, const int* __restrict__ seq_cls, int H, int pool, float* __restrict__ out,
                                                         int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part = reinterpret_cast<float*>(smem);                     // [POOL_WAVES][NJ * 256]
    float* red = part + POOL_WAVES * NJ * 256;                        // [POOL_WAVES]
    constexpr int HP = NJ * 256;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t off = seq_off[b];
    const int nk = seq_nk[b];
    float4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int t_begin = pool == KR_POOL_CLS ? (wave == 0 ? seq_cls[b] : 1 << 30) : wave;
    const int t_end = pool == KR_POOL_CLS ? (wave == 0 ? seq_cls[b] + 1 : 0) : nk;
    // PU of the wave's tokens per step: all their loads are issued before the first add; the adds keep the order t, t + 16, t + 32, ... so the result does
    // not depend on the unrolling
    constexpr int PU = NJ <= 4 ? 4 : 2;       // 16 waves per block: 128 registers per lane
    for (int t = t_begin; t < t_end; t += POOL_WAVES * PU) {
        ushort4 hi[PU][NJ]; unsigned int lo[PU][NJ];
#pragma unroll
        for (int u = 0; u < PU; ++u)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int i = lane * 4 + j * 256;
                if (i < H && t + POOL_WAVES * u < t_end) {
                    hi[u][j] = *reinterpret_cast<const ushort4*>(xb + (off + t + POOL_WAVES * u) * H + i);
                    lo[u][j] = xlo ? *reinterpret_cast<const unsigned int*>(xlo + (off + t + POOL_WAVES * u) * H + i) : 0x80808080u;
                }
            }
#pragma unroll
        for (int u = 0; u < PU; ++u)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int i = lane * 4 + j * 256;
                if (i < H && t + POOL_WAVES * u < t_end) {
                    acc[j].x += lo_decode_final(lo[u][j] & 0xffu, ET::to_f32(hi[u][j].x)); acc[j].y += lo_decode_final((lo[u][j] >> 8) & 0xffu, ET::to_f32(hi[u][j].y));
                    acc[j].z += lo_decode_final((lo[u][j] >> 16) & 0xffu, ET::to_f32(hi[u][j].z)); acc[j].w += lo_decode_final(lo[u][j] >> 24, ET::to_f32(hi[u][j].w));
                }
            }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = lane * 4 + j * 256;
        if (i < H) *reinterpret_cast<float4*>(&part[wave * HP + i]) = acc[j];
    }
    __syncthreads();
    constexpr int VJ = (NJ * 256 + POOL_WAVES * 64 - 1) / (POOL_WAVES * 64);      // elements per thread of the pooled row
    float v[VJ];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < VJ; ++j) {
        const int i = tid + j * POOL_WAVES * 64;
        v[j] = 0.f;
        if (i < H) {
            float s = part[i];
#pragma unroll
            for (int w = 1; w < POOL_WAVES; ++w) s += part[w * HP + i];
            v[j] = pool == KR_POOL_CLS ? s : s / (float)nk;   // nk == 0 -> 0/0 = NaN like average_pool (encoders.py:56-58)
            ss += v[j] * v[j];
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float n2 = red[0];
#pragma unroll
    for (int w = 1; w < POOL_WAVES; ++w) n2 += red[w];
    const float nrm = sqrtf(n2);
    // A sequence WITH attended tokens whose pooled vector is not finite: an activation left the 16-bit operand range upstream (f16: |x| > 65504 becomes
    // inf, the next LayerNorm row NaN) or the weights hold NaN / Inf.  Recorded in the sticky error word (bit 1) and reported like an out-of-vocabulary
    // token id (kr_encoder_check: KR_ERANGE) instead of being returned as an embedding.  nk == 0 is the reference's own NaN (average_pool of nothing).
    if (tid == 0 && nk > 0 && !(nrm < INFINITY)) atomicOr(err, 2);
    const float den = fmaxf(nrm, 1e-12f);   // F.normalize eps; NaN norm stays NaN (fmaxf would drop it)
#pragma unroll
    for (int j = 0; j < VJ; ++j) {
        const int i = tid + j * POOL_WAVES * 64;
        if (i < H) out[(int64_t)b * H + i] = (nrm == nrm) ? v[j] / den : NAN;
    }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
template <class P>
static int dmalloc(P** p, size_t bytes) {
    KR_HIP(hipMalloc(reinterpret_cast<void**>(p), bytes));
    return 0;
}

static void free_ws(Encoder* e) {
    if (!e->graphs.empty()) { (void)hipDeviceSynchronize(); drop_graphs(e); }   // captured kernels hold workspace pointers
    void* ptrs[] = {e->d_ids, e->d_mask, e->d_tt, e->tok_type, e->seq_off, e->seq_nk, e->seq_nq, e->seq_cls, e->seq_has0, e->tok_id, e->tok_pos, e->xlo, e->y, e->out,
                    e->xb, e->q, e->k, e->vT, e->ctx, e->h, e->c_ctx, e->c_xb, e->c_y, e->c_h, e->c_xlo, e->c_off, e->c_nk, e->c_cls, e->d_B};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    e->c_ctx = e->c_xb = e->c_y = e->c_h = nullptr; e->c_xlo = nullptr; e->c_off = e->c_nk = e->c_cls = e->d_B = nullptr;
    e->d_ids = e->d_mask = e->d_tt = nullptr; e->tok_type = nullptr; e->seq_off = e->seq_nk = e->seq_nq = e->seq_cls = e->seq_has0 = nullptr; e->tok_id = e->tok_pos = nullptr;
    e->out = nullptr; e->xlo = nullptr; e->y = e->xb = e->q = e->k = e->vT = e->ctx = e->h = nullptr;
    e->capT = 0; e->capB = 0; e->capBS = 0;
}

static int ensure_ws(Encoder* e, int B, int S) {
    const int64_t maxT = (int64_t)B * (S + 8);
    const int H = e->cfg.hidden, FF = e->cfg.intermediate;
    if (!e->d_T) {
        KR_TRY(dmalloc(&e->d_T, sizeof(int))); KR_TRY(dmalloc(&e->d_err, sizeof(int)));
        KR_HIP(hipMemset(e->d_err, 0, sizeof(int)));
        KR_HIP(hipHostMalloc(reinterpret_cast<void**>(&e->h_err), sizeof(int), hipHostMallocDefault));
        *e->h_err = 0;
        KR_HIP(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
    }
    if (maxT <= e->capT && B <= e->capB && (int64_t)B * S <= e->capBS) return 0;
    if (maxT > (int64_t)1 << 24) return fail(KR_EINVAL, "batch of %lld tokens: at most 2^24 per forward (32-bit offsets inside the attention kernels)", (long long)maxT);
    free_ws(e);
    const int64_t capT = round_up(maxT, 256), capB = B, capBS = (int64_t)B * S;   // multiple of the 256-token tile: see k_proj
    KR_TRY(dmalloc(&e->d_ids, capBS * 8)); KR_TRY(dmalloc(&e->d_mask, capBS * 8)); KR_TRY(dmalloc(&e->d_tt, capBS * 8));
    KR_TRY(dmalloc(&e->seq_off, capB * 4)); KR_TRY(dmalloc(&e->seq_nk, capB * 4)); KR_TRY(dmalloc(&e->seq_nq, capB * 4)); KR_TRY(dmalloc(&e->seq_cls, capB * 4)); KR_TRY(dmalloc(&e->seq_has0, capB * 4));
    KR_TRY(dmalloc(&e->tok_id, capT * 4)); KR_TRY(dmalloc(&e->tok_pos, capT * 4)); KR_TRY(dmalloc(&e->tok_type, capT * 4));
    KR_TRY(dmalloc(&e->xlo, capT * H)); KR_TRY(dmalloc(&e->y, capT * H * 2)); KR_TRY(dmalloc(&e->out, (size_t)capB * H * 4));
    KR_TRY(dmalloc(&e->xb, capT * H * 2)); KR_TRY(dmalloc(&e->q, capT * H * 2));
    KR_TRY(dmalloc(&e->k, (capT + 64) * H * 2));    // 64 rows of slack: k_attn_dma reads whole 64-key chunks (the rows past a sequence are masked)
    KR_HIP(hipMemset(e->k, 0, (size_t)(capT + 64) * H * 2));
    e->ldv = capT + 64;   // slack: the last key tile of the last sequence may read up to 43 columns past T
    KR_TRY(dmalloc(&e->vT, (size_t)H * e->ldv * 2));
    KR_HIP(hipMemset(e->vT, 0, (size_t)H * e->ldv * 2));
    KR_TRY(dmalloc(&e->ctx, capT * H * 2)); KR_TRY(dmalloc(&e->h, capT * (FF + e->h_pad) * 2));
    if (e->cls_shortcut) {
        const int64_t capC = round_up(capB, 256);                 // token-indexed buffers come in multiples of the 256-row tile (see k_proj)
        KR_TRY(dmalloc(&e->c_ctx, capC * H * 2)); KR_TRY(dmalloc(&e->c_xb, capC * H * 2)); KR_TRY(dmalloc(&e->c_y, capC * H * 2)); KR_TRY(dmalloc(&e->c_xlo, capC * H));
        KR_TRY(dmalloc(&e->c_h, capC * (FF + e->h_pad) * 2));
        KR_HIP(hipMemset(e->c_ctx, 0, (size_t)capC * H * 2)); KR_HIP(hipMemset(e->c_xb, 0, (size_t)capC * H * 2)); KR_HIP(hipMemset(e->c_xlo, 0x80, (size_t)capC * H));
        KR_TRY(dmalloc(&e->c_off, capB * 4)); KR_TRY(dmalloc(&e->c_nk, capB * 4)); KR_TRY(dmalloc(&e->c_cls, capB * 4)); KR_TRY(dmalloc(&e->d_B, sizeof(int)));
    }
    e->capT = capT; e->capB = (int)capB; e->capBS = capBS;
    return 0;
}

// tensor slot ids: 0..4 embeddings, then 16 per layer
enum { T_WORD = 0, T_POS, T_TYPE, T_ELNG, T_ELNB, T_LAYER0 };
enum { L_QW = 0, L_QB, L_KW, L_KB, L_VW, L_VB, L_OW, L_OB, L_LN1G, L_LN1B, L_IW, L_IB, L_FW, L_FB, L_LN2G, L_LN2B, L_COUNT };

static int parse_name(const Encoder* e, const char* name, int& slot, int64_t& numel) {
    std::string s(name);
    size_t p = s.find("embeddings.");
    size_t pl = s.find("encoder.layer.");
    const int64_t H = e->cfg.hidden, FF = e->cfg.intermediate;
    if (s.find("position_ids") != std::string::npos || s.find("pooler.") != std::string::npos) { slot = -1; return 0; }
    if (pl != std::string::npos) {
        const char* c = s.c_str() + pl + strlen("encoder.layer.");
        char* end = nullptr;
        const long l = strtol(c, &end, 10);
        if (end == c || *end != '.' || l < 0 || l >= e->cfg.layers) return fail(KR_EINVAL, "bad layer index in '%s'", name);
        const std::string r(end + 1);
        static const struct { const char* n; int id; } tbl[] = {
            {"attention.self.query.weight", L_QW}, {"attention.self.query.bias", L_QB}, {"attention.self.key.weight", L_KW},
            {"attention.self.key.bias", L_KB}, {"attention.self.value.weight", L_VW}, {"attention.self.value.bias", L_VB},
            {"attention.output.dense.weight", L_OW}, {"attention.output.dense.bias", L_OB},
            {"attention.output.LayerNorm.weight", L_LN1G}, {"attention.output.LayerNorm.bias", L_LN1B},
            {"intermediate.dense.weight", L_IW}, {"intermediate.dense.bias", L_IB}, {"output.dense.weight", L_FW}, {"output.dense.bias", L_FB},
            {"output.LayerNorm.weight", L_LN2G}, {"output.LayerNorm.bias", L_LN2B}};
        for (const auto& t : tbl)
            if (r == t.n) {
                slot = T_LAYER0 + (int)l * L_COUNT + t.id;
                switch (t.id) {
                    case L_QW: case L_KW: case L_VW: case L_OW: numel = H * H; break;
                    case L_IW: case L_FW: numel = H * FF; break;
                    case L_IB: numel = FF; break;
                    default: numel = H;
                }
                return 0;
            }
        return fail(KR_EINVAL, "unknown layer tensor '%s'", name);
    }
    if (p != std::string::npos) {
        const std::string r = s.substr(p + strlen("embeddings."));
        if (r == "word_embeddings.weight") { slot = T_WORD; numel = (int64_t)e->cfg.vocab * H; return 0; }
        if (r == "position_embeddings.weight") { slot = T_POS; numel = (int64_t)e->cfg.max_pos * H; return 0; }
        if (r == "token_type_embeddings.weight") { slot = T_TYPE; numel = (int64_t)e->cfg.type_vocab * H; return 0; }
        if (r == "LayerNorm.weight") { slot = T_ELNG; numel = H; return 0; }
        if (r == "LayerNorm.bias") { slot = T_ELNB; numel = H; return 0; }
    }
    return fail(KR_EINVAL, "unknown tensor name '%s'", name);
}

template <int HPB>
static int launch_attn(const Encoder* e, int B, int cap, int nqt, hipStream_t st) {
    const int H = e->cfg.hidden, heads = e->cfg.heads;
    const int kchunk = cap < 128 ? cap : 128;            // keys staged at a time: K 16 KiB + V^T 16.5 KiB per head -> 49 KiB per block, 3 blocks per CU for any S
    const int lds = HPB * (kchunk * 128 + 64 * (kchunk * 2 + 8)) + 4 * 4096;
    static int attr_lds_dev[64] = {};   // per device: function attributes belong to the device's code object instance
    int& attr_lds = attr_lds_dev[e->device & 63];
    if (lds > attr_lds) {
        KR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_lds<HPB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_lds = lds;
    }
    const int qgroups = (nqt + (4 / HPB) - 1) / (4 / HPB);   // blocks per (sequence, head group): 4 / HPB q-tiles each
    hipLaunchKernelGGL((k_attn_lds<HPB>), dim3((unsigned)((heads + HPB - 1) / HPB), (unsigned)B, (unsigned)qgroups), dim3(256), lds, st, e->q, e->k, e->vT,
                       e->ldv, e->seq_off, e->seq_nk, e->seq_nq, H, heads, kchunk, e->ctx);
    return 0;
}

static int launch_attn_dma(const Encoder* e, int B, int nqt, hipStream_t st) {
    const int qgroups = (nqt + ADMA_QT - 1) / ADMA_QT;       // blocks per (sequence, head): 8 q-tiles each
    const int64_t pairs = (int64_t)B * e->cfg.heads;
    const int64_t blocks = (pairs + 7) / 8 * 8 * qgroups;    // whole groups of 8 pairs (see k_attn_dma)
    hipLaunchKernelGGL(k_attn_dma, dim3((unsigned)blocks), dim3(ADMA_THREADS), ADMA_LDS, st, e->q, e->k, e->vT, e->ldv,
                       e->seq_off, e->seq_nk, e->seq_nq, e->cfg.hidden, e->capT, e->ctx, e->cfg.heads, B, qgroups);
    return 0;
}

static int set_lds_once(const void* kern, int lds, int device);
template <class Shape, int STAGES, bool NT>
static int launch_proj_shape_nt(int epi, const ProjArgs& a, int blocks, int device, hipStream_t st) {
    constexpr int lds = STAGES * Shape::STAGE_BYTES + Shape::NWAVE * EPI_STAGE_BYTES;   // 160 KiB for the 256x256 tile: the whole LDS of a CU
    // function attributes belong to the device's code-object instance: once per (kernel, device)
    auto go = [&](auto kern, int) -> int {
        KR_TRY(set_lds_once(reinterpret_cast<const void*>(kern), lds, device));
        int repeat = 1;
#ifdef KR_STAMP
        { const int sl = (epi % 3) + (a.K > 1024 ? 2 : 0); (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(kr_stamp_slot), &sl, sizeof(int), 0, hipMemcpyHostToDevice, st); }
        { const char* r = getenv("KIRAG_AMD_DEBUG_REPEAT"); if (r) repeat = atoi(r); }   // diagnostic: the same launch again (warm instruction cache?)
#endif
        for (int rep = 0; rep < repeat; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(Shape::NTHREADS), lds, st, a);
        return 0;
    };
    const bool ant = a.ant && Shape::BM == 256;   // the non-temporal activation operand exists for the ping-pong loop only
    switch (epi) {
        case EPI_QKV: return go(&k_proj<EPI_QKV, Shape, STAGES, NT>, 0);
        case EPI_DENSE: return ant ? go(&k_proj<EPI_DENSE, Shape, STAGES, NT, true>, 1) : go(&k_proj<EPI_DENSE, Shape, STAGES, NT>, 2);
        case EPI_GELU: return go(&k_proj<EPI_GELU, Shape, STAGES, NT>, 3);
        default: return fail(KR_EINVAL, "projection epilogue %d is not built into this library", epi);
    }
}
template <class Shape, int STAGES>
static int launch_proj_shape(int epi, const ProjArgs& a, int blocks, int device, hipStream_t st) {
    return a.nt ? launch_proj_shape_nt<Shape, STAGES, true>(epi, a, blocks, device, st) : launch_proj_shape_nt<Shape, STAGES, false>(epi, a, blocks, device, st);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device), safe against concurrent first launches from several host threads (a query
// encoder and a corpus encoder on one device: ADVICE r04) and with the outcome cached, so that a failing attribute is reported by every launch
static int set_lds_once(const void* kern, int lds, int device) {
    struct Ent { const void* k; int dev; hipError_t rc; };
    static std::mutex mu;
    static std::vector<Ent> done;
    std::lock_guard<std::mutex> lock(mu);
    for (const Ent& x : done)
        if (x.k == kern && x.dev == device) { if (x.rc != hipSuccess) return fail(KR_EHIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(x.rc)); return 0; }
    const hipError_t rc = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    done.push_back({kern, device, rc});
    if (rc != hipSuccess) return fail(KR_EHIP, "hipFuncSetAttribute failed: %s", hipGetErrorString(rc));
    return 0;
}

// tile shape per launch: when the 256x256 tiling has fewer tiles than ~5/8 of the CUs (small batches: the reference's per_gpu_batch_size 4-8, the KiRAG
// loop's triple batches, a 1/8 slice of a query batch) the 128x128 tiling gives 4x the parallelism at a quarter of the per-tile latency
static int launch_proj(int epi, const ProjArgs& a_in, int64_t max_tokens, const Encoder* e, hipStream_t st) {
    const Knobs& kn = e->kn;
    const int num_cu = e->num_cu, device = e->device;
    ProjArgs a = a_in;
    if (a.ldx == 0) a.ldx = a.K;
    if (a.ldo == 0) a.ldo = a.F;
    a.pw = kn.pw;
    a.epi_prio = kn.epi_prio;
    // store policy by output size (see ProjArgs::nt)
    a.nt = kn.store_nt >= 0 ? kn.store_nt : (max_tokens * (int64_t)a.F * 2 > ((int64_t)96 << 20) ? 1 : 0);
    // FF2 (K = FF > H): its activation operand h is a once-through stream four times the size of every other activation (256 MiB at 32 k tokens); loaded
    // non-temporally it leaves the L2 / Infinity Cache to the weights and to the residual stream the LayerNorm behind it reads: -0.7 % forward time at
    // 1000 x 32 tokens, neutral elsewhere (profiles/r04/tried_nt_activations.txt; outputs bit-identical).  KIRAG_AMD_NT_H=0 switches it off (A/B).
    a.ant = (a.K > a.H && kn.nt_h) ? 1 : 0;
    const int64_t big_tiles = ((max_tokens + 255) / 256) * ((a.F + 255) / 256);
    const int64_t small_tiles = ((max_tokens + 127) / 128) * ((a.F + 127) / 128);
    const int force = kn.force_tile, ratio8 = kn.ratio8;
    const bool small = force == 128 || (force != 256 && big_tiles * 8 < (int64_t)num_cu * ratio8);   // measured crossover: ~5/8 of the CUs busy with 256x256 tiles
    // a handful of token rows: the skinny loop, one tile per block (latency chain: the operand stream of the launch spread over as many CUs as it has
    // tiles).  32 x 32 tiles while there is at most one per CU, 64 x 64 (four multiplying waves share every staged K-tile: half the L2 -> LDS bytes per
    // output element) up to four 32 x 32 tiles' worth per CU — the crossover against the 128x128 producer / consumer loop measured in round 1.
    // KIRAG_AMD_PROJ_TILE = 32 / 64 forces one of the two.
    const int64_t skinny_tiles = ((max_tokens + 31) / 32) * (a.F / 32);
    if (force == 32 || force == 64 || (force == 0 && skinny_tiles <= 4 * (int64_t)num_cu && (max_tokens + 31) / 32 <= 65535)) {
        const bool one = force == 32 || (force == 0 && skinny_tiles <= num_cu);
        auto launch = [&](auto shape_tag, auto kern) -> int {
            constexpr int WM = decltype(shape_tag)::value, RING = WM == 1 ? 16 : 8;
            using G = SkinnyGeom<WM, WM>;
            constexpr int lds = RING * G::STAGE;
            const dim3 grid((unsigned)(a.F / (32 * WM)), (unsigned)((max_tokens + 32 * WM - 1) / (32 * WM)));
            KR_TRY(set_lds_once(reinterpret_cast<const void*>(kern), lds, device));
            hipLaunchKernelGGL(kern, grid, dim3(G::THREADS), lds, st, a);
            return 0;
        };
        auto pick = [&](auto shape_tag) -> int {
            constexpr int WM = decltype(shape_tag)::value, RING = WM == 1 ? 16 : 8;
            switch (epi) {
                case EPI_QKV: return launch(shape_tag, &k_proj_skinny<EPI_QKV, RING, WM, WM>);
                case EPI_DENSE: return launch(shape_tag, &k_proj_skinny<EPI_DENSE, RING, WM, WM>);
                case EPI_GELU: return launch(shape_tag, &k_proj_skinny<EPI_GELU, RING, WM, WM>);
                default: return fail(KR_EINVAL, "projection epilogue %d is not built into this library", epi);
            }
        };
        return one ? pick(std::integral_constant<int, 1>{}) : pick(std::integral_constant<int, 2>{});
    }
    if (!small) return launch_proj_shape<ShapeBig, 2>(epi, a, num_cu, device, st);
    // producer/consumer loop (one persistent block per CU): at most one tile per CU, or more than two (measured: 600 tiles -7 % vs the streaming loop
    // with two blocks per CU; between one and two tiles per CU the two co-resident streaming blocks quantise better).  130 forces it, 128 the streaming loop
    if (((small_tiles <= num_cu || small_tiles > 2 * num_cu) && force != 128) || force == 130) {
        constexpr int lds = SPLIT_RING * ShapeSplit::STAGE_BYTES + 4 * EPI_STAGE_BYTES;
        auto go = [&](auto kern) -> int {
            KR_TRY(set_lds_once(reinterpret_cast<const void*>(kern), lds, device));
            hipLaunchKernelGGL(kern, dim3(num_cu), dim3(SPLIT_THREADS), lds, st, a);
            return 0;
        };
        switch (epi) {
            case EPI_QKV: return go(&k_proj_split<EPI_QKV>);
            case EPI_DENSE: return go(&k_proj_split<EPI_DENSE>);
            case EPI_GELU: return go(&k_proj_split<EPI_GELU>);
            default: return fail(KR_EINVAL, "projection epilogue %d is not built into this library", epi);
        }
    }
    return launch_proj_shape<ShapeSmall, 2>(epi, a, 2 * num_cu, device, st);
}

// ---------------------------------------------------------------------------------------------------------
// entry points of this operand type (C linkage lives in encoder_api.hip, which dispatches on the handle's type)
// ---------------------------------------------------------------------------------------------------------
void enc_destroy(void* h);

int enc_create(const kr_bert_cfg* cfg, int device, int residual_lo, void** out) {
    if (!out || !cfg) return fail(KR_EINVAL, "NULL argument");
    *out = nullptr;
    if (cfg->hidden <= 0 || cfg->hidden % 128 != 0 || cfg->hidden > 2048) return fail(KR_EINVAL, "hidden=%d unsupported (multiple of 128, <= 2048)", cfg->hidden);
    if (cfg->heads <= 0 || cfg->hidden != cfg->heads * 64) return fail(KR_EINVAL, "hidden/heads must be 64 (got %d/%d)", cfg->hidden, cfg->heads);
    if (cfg->intermediate <= 0 || cfg->intermediate % 128 != 0) return fail(KR_EINVAL, "intermediate=%d must be a multiple of 128", cfg->intermediate);
    if (cfg->layers <= 0 || cfg->vocab <= 0 || cfg->max_pos <= 0 || cfg->type_vocab <= 0) return fail(KR_EINVAL, "bad BERT config");
    KR_TRY(select_device(device));
    Encoder* e = new Encoder();
    e->cfg = *cfg; e->device = device;
    e->use_lo = residual_lo != 0;
    { const char* v = getenv("KIRAG_AMD_CLS_FULL"); e->cls_shortcut = !(v && atoi(v) != 0); }
    { const char* v = getenv("KIRAG_AMD_GRAPH"); e->graphs_off = !(v && atoi(v) != 0); }   // opt-in: measured SLOWER than eager launches on ROCm 7.2 (see run_forward)
    { const char* v = getenv("KIRAG_AMD_HPAD"); e->h_pad = v ? (atoi(v) / 8) * 8 : 0; }
    { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, device) == hipSuccess && p.multiProcessorCount > 0) e->num_cu = (p.multiProcessorCount / 8) * 8; }
    e->num_cu_all = e->num_cu;
    e->L.resize(cfg->layers);
    e->got.assign(T_LAYER0 + (size_t)cfg->layers * L_COUNT, 0);
    const size_t H = cfg->hidden, FF = cfg->intermediate;
    int rc = 0;
    auto A = [&](auto** p, size_t bytes) { if (!rc) rc = dmalloc(p, bytes); };
    A(&e->word, (size_t)cfg->vocab * H * 4); A(&e->pos, (size_t)cfg->max_pos * H * 4); A(&e->type, (size_t)cfg->type_vocab * H * 4);
    A(&e->elng, H * 4); A(&e->elnb, H * 4);
    for (auto& l : e->L) {
        A(&l.wqkv, 3 * H * H * 2); A(&l.wo, H * H * 2); A(&l.w1, FF * H * 2); A(&l.w2, H * FF * 2);
        A(&l.bqkv, 3 * H * 4); A(&l.bo, H * 4); A(&l.bo_eff, H * 4); A(&l.b1, FF * 4); A(&l.b2, H * 4);
        A(&l.ln1g, H * 4); A(&l.ln1b, H * 4); A(&l.ln2g, H * 4); A(&l.ln2b, H * 4);
    }
    if (rc) { enc_destroy(e); return rc; }
    *out = e;
    return 0;
}

void enc_destroy(void* h) {
    if (!h) return;
    Encoder* e = reinterpret_cast<Encoder*>(h);
    (void)hipSetDevice(e->device);
    free_ws(e);
    if (e->h_err) (void)hipHostFree(e->h_err);
    if (e->ev_done) (void)hipEventDestroy(e->ev_done);
    if (e->ev_in) (void)hipEventDestroy(e->ev_in);
    if (e->ev_out) (void)hipEventDestroy(e->ev_out);
    if (e->gstream) (void)hipStreamDestroy(e->gstream);
    void* ptrs[] = {e->word, e->pos, e->type, e->elng, e->elnb, e->stage, e->d_T, e->d_err};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& l : e->L) {
        void* lp[] = {l.wqkv, l.wo, l.w1, l.w2, l.bqkv, l.bo, l.bo_eff, l.b1, l.b2, l.ln1g, l.ln1b, l.ln2g, l.ln2b};
        for (void* p : lp) if (p) (void)hipFree(p);
    }
    delete e;
}

int enc_load_weight(void* h, const char* hf_name, const float* data, int64_t numel) {
    if (!h || !hf_name || !data) return fail(KR_EINVAL, "NULL argument");
    Encoder* e = reinterpret_cast<Encoder*>(h);
    KR_TRY(select_device(e->device));
    int slot = -1; int64_t want = 0;
    KR_TRY(parse_name(e, hf_name, slot, want));
    if (slot < 0) return 0;   // pooler.* / position_ids: not used by the encoders (encoders.py:74,115 take last_hidden_state)
    if (numel != want) return fail(KR_EINVAL, "tensor '%s' has %lld elements, expected %lld", hf_name, (long long)numel, (long long)want);
    if ((size_t)numel > e->stage_elems) {
        if (e->stage) (void)hipFree(e->stage);
        e->stage = nullptr; e->stage_elems = 0;
        KR_TRY(dmalloc(&e->stage, (size_t)numel * 4));
        e->stage_elems = (size_t)numel;
    }
    KR_HIP(hipMemcpy(e->stage, data, (size_t)numel * 4, hipMemcpyDefault));
    const int64_t H = e->cfg.hidden;
    const unsigned grid = (unsigned)((numel + 255) / 256);
    auto to_bf16 = [&](uint16_t* dst, float scale) { hipLaunchKernelGGL(k_f32_to_bf16, dim3(grid), dim3(256), 0, 0, e->stage, dst, numel, scale); };
    auto to_f32 = [&](float* dst, float scale) { hipLaunchKernelGGL(k_scale_copy, dim3(grid), dim3(256), 0, 0, e->stage, dst, numel, scale); };
    if (slot < T_LAYER0) {
        float* dst[] = {e->word, e->pos, e->type, e->elng, e->elnb};
        to_f32(dst[slot], 1.f);
    } else {
        LayerW& l = e->L[(slot - T_LAYER0) / L_COUNT];
        switch ((slot - T_LAYER0) % L_COUNT) {
            // log2(e) / sqrt(d_h) is folded into the query projection: the attention scores come out in log2 units and the softmax is a bare exp2
            case L_QW: to_bf16(l.wqkv, 0.125f * 1.4426950408889634f); break;
            case L_QB: to_f32(l.bqkv, 0.125f * 1.4426950408889634f); break;
            case L_KW: to_bf16(l.wqkv + H * H, 1.f); break;
            case L_KB: to_f32(l.bqkv + H, 1.f); break;
            case L_VW: to_bf16(l.wqkv + 2 * H * H, 1.f); break;
            case L_VB: to_f32(l.bqkv + 2 * H, 1.f); break;
            case L_OW: to_bf16(l.wo, 1.f); break;
            case L_OB: to_f32(l.bo, 1.f); break;
            case L_LN1G: to_f32(l.ln1g, 1.f); break;
            case L_LN1B: to_f32(l.ln1b, 1.f); break;
            case L_IW: to_bf16(l.w1, 1.f); break;
            case L_IB: to_f32(l.b1, 1.f); break;
            case L_FW: to_bf16(l.w2, 1.f); break;
            case L_FB: to_f32(l.b2, 1.f); break;
            case L_LN2G: to_f32(l.ln2g, 1.f); break;
            case L_LN2B: to_f32(l.ln2b, 1.f); break;
        }
    }
    KR_HIP(hipGetLastError());
    KR_HIP(hipDeviceSynchronize());
    e->got[slot] = 1;
    e->ready = false;
    return 0;
}

int enc_finalize(void* h) {
    if (!h) return fail(KR_EINVAL, "NULL argument");
    Encoder* e = reinterpret_cast<Encoder*>(h);
    for (size_t i = 0; i < e->got.size(); ++i)
        if (!e->got[i]) return fail(KR_ESTATE, "weight slot %zu (layer %d, tensor %d) was never loaded", i,
                                    i < T_LAYER0 ? -1 : (int)((i - T_LAYER0) / L_COUNT), i < T_LAYER0 ? (int)i : (int)((i - T_LAYER0) % L_COUNT));
    if (e->stage) { (void)hipFree(e->stage); e->stage = nullptr; e->stage_elems = 0; }
    KR_TRY(select_device(e->device));
    const int H = e->cfg.hidden;
    for (auto& l : e->L) hipLaunchKernelGGL(k_fold_vbias, dim3((H + 127) / 128), dim3(128), 0, 0, l.wo, l.bo, l.bqkv + 2 * H, l.bo_eff, H);
    KR_HIP(hipGetLastError());
    KR_HIP(hipDeviceSynchronize());
    e->ready = true;
    return 0;
}

// CLS shortcut: row seq_off[b] + seq_cls[b] of ctx / the residual stream (hi, lo) -> row b of the compact buffers; also the compact "sequence" tables
// (one token per sequence) and the row count for the B-row kernels that follow
__global__ __launch_bounds__(256) void k_gather_cls(const uint16_t* __restrict__ ctx, const uint16_t* __restrict__ xb, const uint8_t* __restrict__ xlo,
                                                    const int* __restrict__ seq_off, const int* __restrict__ seq_cls, const int* __restrict__ seq_nk, int H, uint16_t* __restrict__ c_ctx,
                                                    uint16_t* __restrict__ c_xb, uint8_t* __restrict__ c_xlo, int* __restrict__ c_off, int* __restrict__ c_nk,
                                                    int* __restrict__ c_cls, int* __restrict__ d_B) {
    const int b = blockIdx.x;
    const int64_t src = (int64_t)seq_off[b] + seq_cls[b];
    for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
        *reinterpret_cast<uint4*>(c_ctx + (int64_t)b * H + i) = *reinterpret_cast<const uint4*>(ctx + src * H + i);
        *reinterpret_cast<uint4*>(c_xb + (int64_t)b * H + i) = *reinterpret_cast<const uint4*>(xb + src * H + i);
        if (xlo) *reinterpret_cast<uint2*>(c_xlo + (int64_t)b * H + i) = *reinterpret_cast<const uint2*>(xlo + src * H + i);
    }
    if (threadIdx.x == 0) {
        c_off[b] = b; c_nk[b] = seq_nk[b] > 0 ? 1 : 0; c_cls[b] = 0;   // c_nk == 0: an all-masked sequence (its NaN is the reference's own, not an overflow: k_pool)
        if (b == 0) *d_B = (int)gridDim.x;
    }
}

// every kernel of one forward, enqueued on `st` (inputs already in e->d_ids / e->d_mask, result left in e->out)
static int enqueue_forward(Encoder* e, int B, int S, int pool, hipStream_t st, bool has_tt = false, int ragged_total = -1) {
    const int H = e->cfg.hidden, FF = e->cfg.intermediate;
    const float eps = e->cfg.ln_eps;
    e->kn.read();
    const int nqt_max = (S + (pool == KR_POOL_CLS ? 1 : 0) + 31) / 32;             // q-tiles of the longest possible sequence
    const bool long_seq = (nqt_max > 4 && !e->kn.attn_lds) || e->kn.attn_dma;   // KIRAG_AMD_ATTN_DMA=1: the ring kernel for short sequences too (A/B)
            // > 128 tokens: the LDS-DMA attention kernel (KIRAG_AMD_ATTN_LDS=1: A/B against the register-staged one)
    const int align = long_seq ? 8 : 4;                                            // sequence offsets: multiple of 8 tokens so that V^T chunks start 16-B aligned
    if (ragged_total >= 0) {
        // kr_encoder_forward_packed: e->d_ids holds the int32 token list, e->d_mask the int32 lengths, e->d_tt is scratch for the input offsets
        const int* ids32 = reinterpret_cast<const int*>(e->d_ids);
        const int* lens = reinterpret_cast<const int*>(e->d_mask);
        int* in_off = reinterpret_cast<int*>(e->d_tt);
        if (B <= PACK_SMALL_B) {
            hipLaunchKernelGGL(k_rag_small, dim3(1), dim3(1024), 0, st, ids32, lens, B, S, ragged_total, e->cfg.vocab, pool, align, e->seq_nk, e->seq_has0, e->seq_nq,
                               e->seq_off, e->seq_cls, e->d_T, e->tok_id, e->tok_pos, e->tok_type, e->d_err);
        } else {
            hipLaunchKernelGGL(k_rag_scan, dim3(1), dim3(64), 0, st, lens, B, S, ragged_total, pool, align, e->seq_nk, e->seq_has0, e->seq_nq, e->seq_off, e->seq_cls,
                               in_off, e->d_T, e->d_err);
            hipLaunchKernelGGL(k_rag_fill, dim3(B), dim3(64), 0, st, ids32, e->cfg.vocab, align, in_off, e->seq_off, e->seq_nk, e->seq_nq, e->tok_id, e->tok_pos,
                               e->tok_type, e->d_err);
        }
    } else if (B <= PACK_SMALL_B) {
        hipLaunchKernelGGL(k_pack_small, dim3(1), dim3(1024), 0, st, e->d_ids, e->d_mask, has_tt ? e->d_tt : nullptr, B, S, e->cfg.vocab, e->cfg.type_vocab, pool, align,
                           e->seq_nk, e->seq_has0, e->seq_nq, e->seq_off, e->seq_cls, e->d_T, e->tok_id, e->tok_pos, e->tok_type, e->d_err);
    } else {
        hipLaunchKernelGGL(k_seq_len, dim3(B), dim3(64), 0, st, e->d_mask, B, S, e->seq_nk, e->seq_has0);
        hipLaunchKernelGGL(k_seq_scan, dim3(1), dim3(64), 0, st, e->seq_nk, e->seq_has0, B, pool, align, e->seq_nq, e->seq_off, e->seq_cls, e->d_T, e->d_err);
        hipLaunchKernelGGL(k_fill_tokens, dim3(B), dim3(64), 0, st, e->d_ids, e->d_mask, has_tt ? e->d_tt : nullptr, S, e->cfg.vocab, e->cfg.type_vocab, align, e->seq_off,
                           e->seq_nk, e->seq_nq, e->tok_id, e->tok_pos, e->tok_type, e->d_err);
    }
    const int64_t maxT = (int64_t)B * (((S + (pool == KR_POOL_CLS ? 1 : 0)) + align - 1) & ~(align - 1));   // upper bound of the packed token count (each sequence is padded to `align`)
    const unsigned row_grid = (unsigned)((maxT + 3) / 4);
    // LayerNorm streams: y (dead after the kernel) and the low half (next read by the next LayerNorm, ~600 MiB of traffic later) are loaded / stored
    // non-temporally — they do not displace the 16-bit stream the next GEMM reads: -1.3 % forward at 1024 x 128 tokens, -0.1...0.5 % at 1000 x 32; 8 blocks
    // per CU instead of 4: -0.25 % (profiles/r04/tried_ln_policies.txt; outputs bit-identical)
    auto ln_kernel = H <= 512 ? &k_ln16<1, 7> : H <= 1024 ? &k_ln16<2, 7> : &k_ln16<4, 7>;
    auto pool_kernel = H <= 256 ? &k_pool<1> : H <= 512 ? &k_pool<2> : H <= 1024 ? &k_pool<4> : &k_pool<8>;
    const int pool_lds = POOL_WAVES * (H <= 256 ? 1 : H <= 512 ? 2 : H <= 1024 ? 4 : 8) * 256 * 4 + POOL_WAVES * 4;
    unsigned ln_mult = 8u;
    const unsigned ln_grid = std::min(row_grid, (unsigned)e->num_cu_all * ln_mult);   // k_ln is grid-stride (its parameters stay in registers across rows)
    hipLaunchKernelGGL(k_embed_ln, dim3(row_grid), dim3(256), 0, st, e->tok_id, e->tok_pos, e->tok_type, e->d_T, e->word, e->pos, e->type, e->elng, e->elnb, eps, H,
                       e->use_lo ? e->xlo : nullptr, e->xb);
    uint8_t* const lo_rw = e->use_lo ? e->xlo : nullptr;       // low half read / written by the inner LayerNorms
    const bool shortcut = pool == KR_POOL_CLS && e->cls_shortcut && e->c_ctx != nullptr;
    e->last_shortcut = shortcut;
    for (const LayerW& l : e->L) {
        const bool last = (&l == &e->L.back());
        ProjArgs a{};
        a.Tp = e->d_T; a.H = H;
        // q | k | v^T in one GEMM (F = 3H)
        a.W = l.wqkv; a.X = e->xb; a.F = 3 * H; a.K = H; a.bias = l.bqkv; a.out0 = e->q; a.out1 = e->k; a.outT = e->vT; a.ldT = e->ldv; a.ldx = 0; a.ldo = 0;
        KR_TRY(launch_proj(EPI_QKV, a, maxT, e, st));
        {
            const int cap = (int)round_up(S, 32);
            const int nqt = nqt_max;
            if (long_seq) KR_TRY(launch_attn_dma(e, B, nqt, st));
            else if (nqt >= 3) KR_TRY(launch_attn<1>(e, B, cap, nqt, st));
            else if (nqt == 2) KR_TRY(launch_attn<2>(e, B, cap, nqt, st));
            else KR_TRY(launch_attn<4>(e, B, cap, nqt, st));
        }
        if (last && shortcut) {
            // only the CLS row of every sequence is read after this layer: gather those rows and finish the layer on B rows (same kernels, same arithmetic per
            // row: the projection loops are bit-identical for every tiling and LayerNorm is per row, so the embedding does not change by one bit)
            hipLaunchKernelGGL(k_gather_cls, dim3(B), dim3(256), 0, st, e->ctx, e->xb, lo_rw, e->seq_off, e->seq_cls, e->seq_nk, H, e->c_ctx, e->c_xb, e->c_xlo, e->c_off,
                               e->c_nk, e->c_cls, e->d_B);
            uint8_t* const c_lo = e->use_lo ? e->c_xlo : nullptr;
            const unsigned c_ln_grid = std::min((unsigned)((B + 3) / 4), (unsigned)e->num_cu_all * ln_mult);
            a.Tp = e->d_B;
            a.W = l.wo; a.X = e->c_ctx; a.F = H; a.K = H; a.bias = l.bo_eff; a.out0 = e->c_y; a.ldx = 0; a.ldo = 0;
            KR_TRY(launch_proj(EPI_DENSE, a, B, e, st));
            hipLaunchKernelGGL(ln_kernel, dim3(c_ln_grid), dim3(256), 0, st, e->c_y, l.bo_eff, e->d_B, l.ln1g, l.ln1b, eps, H, c_lo, c_lo, e->c_xb);
            a.W = l.w1; a.X = e->c_xb; a.F = FF; a.K = H; a.bias = l.b1; a.out0 = e->c_h; a.ldx = 0; a.ldo = FF + e->h_pad;
            KR_TRY(launch_proj(EPI_GELU, a, B, e, st));
            a.W = l.w2; a.X = e->c_h; a.F = H; a.K = FF; a.bias = l.b2; a.out0 = e->c_y; a.ldx = FF + e->h_pad; a.ldo = 0;
            KR_TRY(launch_proj(EPI_DENSE, a, B, e, st));
            hipLaunchKernelGGL(ln_kernel, dim3(c_ln_grid), dim3(256), 0, st, e->c_y, l.b2, e->d_B, l.ln2g, l.ln2b, eps, H, c_lo, e->c_xlo, e->c_xb);
            KR_TRY(set_lds_once(reinterpret_cast<const void*>(pool_kernel), pool_lds, e->device));
            hipLaunchKernelGGL(pool_kernel, dim3(B), dim3(POOL_WAVES * 64), pool_lds, st, e->c_xb, e->c_xlo, e->c_off, e->c_nk, e->c_cls, H, pool, e->out, e->d_err);
            KR_HIP(hipGetLastError());
            return 0;
        }
        // attention.output.dense + residual -> LayerNorm
        a.W = l.wo; a.X = e->ctx; a.F = H; a.K = H; a.bias = l.bo_eff; a.out0 = e->y; a.ldx = 0; a.ldo = 0;
        KR_TRY(launch_proj(EPI_DENSE, a, maxT, e, st));
        hipLaunchKernelGGL(ln_kernel, dim3(ln_grid), dim3(256), 0, st, e->y, l.bo_eff, e->d_T, l.ln1g, l.ln1b, eps, H, lo_rw, lo_rw, e->xb);
        // intermediate.dense + GELU
        a.W = l.w1; a.X = e->xb; a.F = FF; a.K = H; a.bias = l.b1; a.out0 = e->h; a.ldx = 0; a.ldo = FF + e->h_pad;
        KR_TRY(launch_proj(EPI_GELU, a, maxT, e, st));
        // output.dense + residual -> LayerNorm
        a.W = l.w2; a.X = e->h; a.F = H; a.K = FF; a.bias = l.b2; a.out0 = e->y; a.ldx = FF + e->h_pad; a.ldo = 0;
        KR_TRY(launch_proj(EPI_DENSE, a, maxT, e, st));
        // the LAST LayerNorm always writes the low half: pooling and kr_encoder_last_hidden read the final hidden state with 16 mantissa bits
        hipLaunchKernelGGL(ln_kernel, dim3(ln_grid), dim3(256), 0, st, e->y, l.b2, e->d_T, l.ln2g, l.ln2b, eps, H, lo_rw, last ? e->xlo : lo_rw, e->xb);
    }
    KR_TRY(set_lds_once(reinterpret_cast<const void*>(pool_kernel), pool_lds, e->device));
    hipLaunchKernelGGL(pool_kernel, dim3(B), dim3(POOL_WAVES * 64), pool_lds, st, e->xb, e->xlo, e->seq_off, e->seq_nk, e->seq_cls, H, pool, e->out, e->d_err);
    KR_HIP(hipGetLastError());
    return 0;
}

// Small batches are launch-bound (a 1 x 32-token forward is ~175 launches of 6-13 us).  With KIRAG_AMD_GRAPH=1 (read at kr_encoder_create; OFF by
// default) the kernel sequence of a (B, S, pool) shape is replayed as ONE hipGraph from its second forward on.  Measured on MI355X / ROCm 7.2
// (tools/graph_bench.py, two encoders interleaved in one process, outputs bit-identical): replay is SLOWER than the eager launches, both back to back
// and with a synchronisation after every forward — 1 x 32 tokens 1.40 vs 1.20 ms, 2 x 256 2.14 vs 1.95, 8 x 128 2.31 vs 2.12, 100 x 32 3.42 vs 3.29 —
// about 1.1 us per graph node more than the launch it replaces, so the default stays eager (the asynchronous boundary already lets the host run ahead).  The first forward of a shape runs eagerly (it also sets the function
// attributes, which must not happen during capture); capture and replay use an internal stream (the caller's may be the legacy default stream,
// which cannot be captured), ordered against the caller's stream by events.  Any failure falls back to eager launches for good.
constexpr int64_t GRAPH_MAX_TOKENS = 4096;
static int run_forward(Encoder* e, int B, int S, int pool, hipStream_t st, bool has_tt = false) {
    const int64_t maxT = (int64_t)B * (S + 8);
    if (e->graphs_off || maxT > GRAPH_MAX_TOKENS || has_tt) return enqueue_forward(e, B, S, pool, st, has_tt);
    const uint64_t key = ((uint64_t)B << 32) | ((uint64_t)S << 8) | (uint64_t)pool;
    GraphEntry* ent = nullptr;
    for (auto& g : e->graphs) if (g.key == key) { ent = &g; break; }
    if (!ent) {
        if (e->graphs.size() >= 32) drop_graphs(e);
        e->graphs.push_back(GraphEntry{key, 0, nullptr});
        ent = &e->graphs.back();
    }
    ent->calls++;
    if (ent->calls == 1) return enqueue_forward(e, B, S, pool, st);
    if (!e->gstream) {
        if (hipStreamCreateWithFlags(&e->gstream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); e->graphs_off = true; return enqueue_forward(e, B, S, pool, st); }
    }
    if (!ent->exec) {
        hipGraph_t graph = nullptr;
        bool ok = hipStreamBeginCapture(e->gstream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            const int rc = enqueue_forward(e, B, S, pool, e->gstream);
            const hipError_t ce = hipStreamEndCapture(e->gstream, &graph);
            ok = rc == 0 && ce == hipSuccess && graph != nullptr && hipGraphInstantiate(&ent->exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
        }
        if (!ok) { (void)hipGetLastError(); ent->exec = nullptr; e->graphs_off = true; return enqueue_forward(e, B, S, pool, st); }
    }
    KR_HIP(hipEventRecord(e->ev_in, st));
    KR_HIP(hipStreamWaitEvent(e->gstream, e->ev_in, 0));
    KR_HIP(hipGraphLaunch(ent->exec, e->gstream));
    KR_HIP(hipEventRecord(e->ev_out, e->gstream));
    KR_HIP(hipStreamWaitEvent(st, e->ev_out, 0));
    return 0;
}

// the sticky device error word has been copied to e->h_err and that copy has completed
static int report_token_error(Encoder* e, hipStream_t st) {
    const int w = *e->h_err;
    if (w == 0) return 0;
    *e->h_err = 0;
    KR_HIP(hipMemsetAsync(e->d_err, 0, sizeof(int), st));
    if (w & 1) return fail(KR_EINVAL, "input_ids contain a token id outside [0, %d)", e->cfg.vocab);
    if (w & 4) return fail(KR_EINVAL, "token_type_ids contain a value outside [0, %d)", e->cfg.type_vocab);
    if (w & 8) return fail(KR_EINVAL, "seq_lens hold a length outside [0, S] or do not add up to total_tokens (kr_encoder_forward_packed)");
#ifdef KR_ENC_BUILD_F16
    return fail(KR_ERANGE, "non-finite activations in the forward: a value left the f16 operand range (|x| > 65504) or the weights hold NaN / Inf; "
                           "the embeddings of this batch are not usable (KIRAG_AMD_ENCODER_DTYPE=bf16 has the fp32 exponent range)");
#else
    return fail(KR_ERANGE, "non-finite activations in the forward: the weights (or an overflowing accumulation) hold NaN / Inf; the embeddings of this batch are not usable");
#endif
}

int enc_forward(void* h, const int64_t* input_ids, const int64_t* attention_mask, const int64_t* token_type_ids, int B, int S, int pool, float* out, void* stream) {
    if (!h) return fail(KR_EINVAL, "encoder is NULL");
    Encoder* e = reinterpret_cast<Encoder*>(h);
    if (!e->ready) return fail(KR_ESTATE, "encoder weights incomplete: call kr_encoder_finalize after loading every tensor");
    if (B < 0 || S <= 0 || (B > 0 && (!input_ids || !attention_mask || !out))) return fail(KR_EINVAL, "bad input pointers / shape");
    if (S > e->cfg.max_pos) return fail(KR_EINVAL, "sequence length %d exceeds max_position_embeddings %d", S, e->cfg.max_pos);
    if (pool != KR_POOL_MEAN && pool != KR_POOL_CLS) return fail(KR_EINVAL, "pool must be 0 (mean) or 1 (cls)");
    if (B == 0) return 0;
    if (B > 65535) return fail(KR_EINVAL, "at most 65535 sequences per call");
    KR_TRY(select_device(e->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (e->pending) {
        if (st != e->last_stream) KR_HIP(hipEventSynchronize(e->ev_done));      // the workspace is shared: a forward on another stream waits for the previous one
        if (hipEventQuery(e->ev_done) == hipSuccess) {                          // finished: report its token-id error now (never blocks)
            e->pending = false;
            KR_TRY(report_token_error(e, st));
        }
    }
    KR_TRY(ensure_ws(e, B, S));
    const int H = e->cfg.hidden;
    KR_HIP(hipMemcpyAsync(e->d_ids, input_ids, (size_t)B * S * 8, hipMemcpyDefault, st));
    KR_HIP(hipMemcpyAsync(e->d_mask, attention_mask, (size_t)B * S * 8, hipMemcpyDefault, st));
    if (token_type_ids) KR_HIP(hipMemcpyAsync(e->d_tt, token_type_ids, (size_t)B * S * 8, hipMemcpyDefault, st));
    KR_TRY(run_forward(e, B, S, pool, st, token_type_ids != nullptr));
    KR_HIP(hipMemcpyAsync(out, e->out, (size_t)B * H * 4, hipMemcpyDefault, st));
    e->lastB = B; e->lastS = S; e->last_stream = st;
    KR_HIP(hipMemcpyAsync(e->h_err, e->d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    if (is_device_pointer(out)) {
        // device output: nothing here waits for the GPU.  The token-id error word (the only thing the host would need) travels to pinned memory
        // behind the forward; it is looked at by the next call on this handle or by kr_encoder_check().
        KR_HIP(hipEventRecord(e->ev_done, st));
        e->pending = true;
        return 0;
    }
    KR_HIP(hipStreamSynchronize(st));   // a host pointer is returned: the caller reads it as soon as we return
    e->pending = false;
    return report_token_error(e, st);
}

// kr_encoder_forward_packed: the same forward from the ragged token list (int32 ids of the attended positions + int32 length per sequence)
int enc_forward_packed(void* h, const int32_t* token_ids, const int32_t* seq_lens, int B, int S, int64_t total_tokens, int pool, float* out, void* stream) {
    if (!h) return fail(KR_EINVAL, "encoder is NULL");
    Encoder* e = reinterpret_cast<Encoder*>(h);
    if (!e->ready) return fail(KR_ESTATE, "encoder weights incomplete: call kr_encoder_finalize after loading every tensor");
    if (B < 0 || S <= 0 || total_tokens < 0 || (B > 0 && (!seq_lens || !out)) || (total_tokens > 0 && !token_ids)) return fail(KR_EINVAL, "bad input pointers / shape");
    if (S > e->cfg.max_pos) return fail(KR_EINVAL, "sequence length %d exceeds max_position_embeddings %d", S, e->cfg.max_pos);
    if (total_tokens > (int64_t)B * S) return fail(KR_EINVAL, "total_tokens %lld exceeds B * S = %lld", (long long)total_tokens, (long long)B * S);
    if (pool != KR_POOL_MEAN && pool != KR_POOL_CLS) return fail(KR_EINVAL, "pool must be 0 (mean) or 1 (cls)");
    if (B == 0) return 0;
    if (B > 65535) return fail(KR_EINVAL, "at most 65535 sequences per call");
    KR_TRY(select_device(e->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (e->pending) {
        if (st != e->last_stream) KR_HIP(hipEventSynchronize(e->ev_done));
        if (hipEventQuery(e->ev_done) == hipSuccess) {
            e->pending = false;
            KR_TRY(report_token_error(e, st));
        }
    }
    KR_TRY(ensure_ws(e, B, S));
    const int H = e->cfg.hidden;
    if (total_tokens > 0) KR_HIP(hipMemcpyAsync(e->d_ids, token_ids, (size_t)total_tokens * 4, hipMemcpyDefault, st));
    KR_HIP(hipMemcpyAsync(e->d_mask, seq_lens, (size_t)B * 4, hipMemcpyDefault, st));
    KR_TRY(enqueue_forward(e, B, S, pool, st, false, (int)total_tokens));
    KR_HIP(hipMemcpyAsync(out, e->out, (size_t)B * H * 4, hipMemcpyDefault, st));
    e->lastB = B; e->lastS = S; e->last_stream = st;
    KR_HIP(hipMemcpyAsync(e->h_err, e->d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    if (is_device_pointer(out)) {
        KR_HIP(hipEventRecord(e->ev_done, st));
        e->pending = true;
        return 0;
    }
    KR_HIP(hipStreamSynchronize(st));
    e->pending = false;
    return report_token_error(e, st);
}

int enc_check(void* h) {
    if (!h) return fail(KR_EINVAL, "encoder is NULL");
    Encoder* e = reinterpret_cast<Encoder*>(h);
    if (!e->pending) return 0;
    KR_TRY(select_device(e->device));
    KR_HIP(hipEventSynchronize(e->ev_done));
    e->pending = false;
    return report_token_error(e, e->last_stream);
}

#if defined(KR_STAMP) && !defined(KR_ENC_BUILD_F16)
}  // namespace KR_ENC_NS
}  // namespace kr
extern "C" {
using namespace kr;
int kr_debug_read_fine_enc(unsigned long long* out128) {
    if (hipMemcpyFromSymbol(out128, HIP_SYMBOL(kr_stamp_fine), 128 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[128] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(kr_stamp_fine), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
// diagnostic build only: the projection kernels' copy of the stamp sums
int kr_debug_read_stamps_enc(unsigned long long* out256) {
    if (hipMemcpyFromSymbol(out256, HIP_SYMBOL(kr_stamp_buf), 256 * sizeof(unsigned long long)) != hipSuccess) return -1;
    unsigned long long z[256] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(kr_stamp_buf), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
}  // extern "C"
namespace kr {
namespace KR_ENC_NS {
#endif

// last_hidden_state of the previous forward, un-packed to [B,S,H]; rows of non-attended positions are zero
int enc_last_hidden(void* h, float* out, int B, int S) {
    if (!h || !out) return fail(KR_EINVAL, "NULL argument");
    Encoder* e = reinterpret_cast<Encoder*>(h);
    if (B != e->lastB || S != e->lastS || B == 0) return fail(KR_ESTATE, "no forward of shape [%d,%d] to read back", B, S);
    if (e->last_shortcut)
        return fail(KR_ESTATE, "the last forward pooled the CLS token: its last layer ran on the CLS rows only (create the encoder with KIRAG_AMD_CLS_FULL=1 to keep every row)");
    KR_TRY(select_device(e->device));
    KR_HIP(hipStreamSynchronize(e->last_stream));
    const int H = e->cfg.hidden;
    std::vector<int> off(B), nq(B);
    int T = 0;
    KR_HIP(hipMemcpy(off.data(), e->seq_off, B * 4, hipMemcpyDeviceToHost));
    KR_HIP(hipMemcpy(nq.data(), e->seq_nq, B * 4, hipMemcpyDeviceToHost));
    KR_HIP(hipMemcpy(&T, e->d_T, 4, hipMemcpyDeviceToHost));
    std::vector<int> pos(T);
    std::vector<float> x((size_t)T * H);
    KR_HIP(hipMemcpy(pos.data(), e->tok_pos, (size_t)T * 4, hipMemcpyDeviceToHost));
    {   // the final hidden state is stored as (16-bit hi, 8-bit lo in units of ulp(hi) / 256): see lo_encode
        std::vector<uint16_t> hi((size_t)T * H); std::vector<uint8_t> lo((size_t)T * H);
        KR_HIP(hipMemcpy(hi.data(), e->xb, hi.size() * 2, hipMemcpyDeviceToHost));
        KR_HIP(hipMemcpy(lo.data(), e->xlo, lo.size(), hipMemcpyDeviceToHost));
#ifdef KR_ENC_BUILD_F16
        auto f = [](uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); };
#else
        auto f = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float v; std::memcpy(&v, &u, 4); return v; };
#endif
        for (size_t i = 0; i < x.size(); ++i) {      // the host twin of lo_decode_final
            const float hf = f(hi[i]);
            uint32_t bits; std::memcpy(&bits, &hf, 4);
            if ((bits & 0x7f800000u) != 0x7f800000u) bits += ((uint32_t)lo[i] << LO_SH) - (128u << LO_SH);
            std::memcpy(&x[i], &bits, 4);
        }
    }
    std::vector<float> full((size_t)B * S * H, 0.f);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < nq[b]; ++i)
            std::memcpy(&full[((size_t)b * S + pos[off[b] + i]) * H], &x[(size_t)(off[b] + i) * H], (size_t)H * 4);
    KR_HIP(hipMemcpy(out, full.data(), full.size() * 4, hipMemcpyDefault));
    return 0;
}

}  // namespace KR_ENC_NS
}  // namespace kr
