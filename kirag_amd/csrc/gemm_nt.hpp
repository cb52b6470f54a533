// MFMA "NT" GEMM main loop for gfx950:  C[M,N] = A[M,K] * B[N,K]^T  with A and B K-contiguous 16-bit
// (bf16 or f16) and fp32 accumulation.  One main loop serves the index coarse scan (A = corpus rows,
// B = queries, epilogue = threshold filter) and every encoder projection (A = tokens, B = weight [out,in],
// epilogue = bias / GELU / residual).  The result tile stays in registers and is handed to an epilogue functor.
//
// Two main loops share the staging scheme: gemm_nt_pingpong (256x256 tiles, the product path of the coarse scan and of the large
// projections) and gemm_nt_stream (any tile shape, used for 128x128 tiles when a launch has few tiles).
// Staging (cdna_hip_programming.md §5): BK = 64; both operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// wave-instruction = 8 rows x 128 B) into a ring of K-tiles.  The LDS image of a tile is [rows][8 x 16 B]
// with the 16-B chunk index XOR-swizzled by ((row >> 1) & 7): the DMA destination is lane-linear, so the
// swizzle is applied to the per-lane SOURCE address and again on the ds_read_b128 fragment reads (rule 21).
// ds_read_b128 of a fragment (lane -> row l&31, chunk 2*ks + (l>>5)) is then bank-conflict free.
#pragma once
#include "common.hpp"

#include <type_traits>

namespace kr {

template <int BM_, int BN_, int WM_, int WN_>
struct GemmShape {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int BK = 64;
    static constexpr int NWAVE = WM * WN;
    static constexpr int NTHREADS = NWAVE * 64;
    static constexpr int TM = BM / WM / 32;  // 32x32 MFMA tiles per wave along M
    static constexpr int TN = BN / WN / 32;
    static constexpr int A_BYTES = BM * BK * 2;
    static constexpr int B_BYTES = BN * BK * 2;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr int A_PIECES = BM / 8 / NWAVE;  // 1-KiB DMA pieces per wave per tile
    static constexpr int B_PIECES = BN / 8 / NWAVE;
    static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tile must be a multiple of 32x32");
    static_assert((BM / 8) % NWAVE == 0 && (BN / 8) % NWAVE == 0, "DMA pieces must divide over the waves");
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// Accumulator view handed to epilogues: element (mi, ni, r) of lane `lane` is
//   row = m_wave + mi*32 + (r & 3) + 8*(r >> 2) + 4*(lane >> 5),   col = n_wave + ni*32 + (lane & 31).
template <class Shape>
struct AccTile {
    f32x16 v[Shape::TM][Shape::TN];
    int m_wave, n_wave, lane;
#ifdef KR_STAMP
    unsigned long long fine[6] = {0, 0, 0, 0, 0, 0}, fine_prev = 0;
    __device__ __forceinline__ void stamp(int step) { const unsigned long long now = __builtin_amdgcn_s_memtime(); fine[step] += now - fine_prev; fine_prev = now; }
#endif
    __device__ __forceinline__ int row(int mi, int r) const { return m_wave + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
    __device__ __forceinline__ int col(int ni) const { return n_wave + ni * 32 + (lane & 31); }
};

// =====================================================================================================================
// streaming main loop: PERSISTENT blocks streaming a sequence of output tiles through a 2..4-stage LDS ring.
//   * the LDS-DMA of K-tile g+2 is issued while K-tile g is multiplied: two tiles stay in flight across the single
//     raw s_barrier per K-tile, retired by a COUNTED s_waitcnt vmcnt (never 0 in steady state) — cdna_hip_programming.md
//     "Pipelining across barriers"; the ring runs across output-tile boundaries, so a tile's first K-tiles are already
//     landing while the previous tile's epilogue runs
//   * iteration g:  wait own DMA(g) -> s_barrier (RAW: every wave's pieces of tile g landed; WAR: every wave is done reading
//     buffer (g-1)%3) -> issue DMA(g+2) into buffer (g+2)%3 == (g-1)%3 -> multiply buffer g%3
//   * MFMA fragments are double-buffered across the 16-deep k-steps so ds_read latency sits under the previous k-step's MFMAs
// Block ids are virtual: block b takes v = b, b+G, b+2G, ... (G = gridDim.x, a multiple of 8) and xcd_chunk_map turns v into
// a "natural" tile index such that each XCD walks a contiguous run of the natural order.
// =====================================================================================================================

// bijective: v in [0,total) -> natural index; XCD x (= v % 8) owns the contiguous natural range starting at x*q + min(x,r)
__device__ __forceinline__ int64_t xcd_chunk_map(int64_t v, int64_t total) {
    const int64_t q = total >> 3, r = total & 7;
    const int64_t x = v & 7, j = v >> 3;
    return x * q + (x < r ? x : r) + j;
}

// exact n / d and n % d for n < 2^24, 0 < d (fp32 reciprocal + one correction step): ~10 VALU.  The tile coordinates are recomputed at
// every output-tile boundary by every wave (three times: two DMA cursors + the epilogue); a 64-bit integer division costs ~150 VALU
// instructions each and sat in the boundary's critical path (PMC: 900 VALU per tile per wave, of which the epilogue itself is 211).
__device__ __forceinline__ void fast_divmod(uint32_t n, uint32_t d, uint32_t& q, uint32_t& r) {
    q = (uint32_t)((float)n * __builtin_amdgcn_rcpf((float)d));
    r = n - q * d;
    if ((int32_t)r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}
// exact n / d and n % d for n < 2^53 (fp64 reciprocal + one correction step)
__device__ __forceinline__ void fast_divmod64(uint64_t n, uint32_t d, uint64_t& q, uint32_t& r) {
    q = (uint64_t)((double)n * (1.0 / (double)d));
    int64_t rr = (int64_t)(n - q * (uint64_t)d);
    if (rr < 0) { --q; rr += d; }
    else if (rr >= (int64_t)d) { ++q; rr -= d; }
    r = (uint32_t)rr;
}

// natural index -> (tm, tn) such that 8 consecutive tn of one tm-run are adjacent: groups of (up to) 8 tn, tm walks inside a group.
// Used by the encoder projections only: at most 65535 x 512 tokens = 131k token tiles x 16 feature tiles = 2^21 tiles < 2^24.
__device__ __forceinline__ void patch_coord(int64_t n, int64_t tm_count, int64_t tn_count, int64_t& tm, int64_t& tn, uint32_t pw = 8u) {
    uint32_t g, rem, q, r;                      // pw = feature tiles per patch (8: the 32 CUs of an XCD hold 4 token tiles x 8 feature tiles at a time)
    fast_divmod((uint32_t)n, pw * (uint32_t)tm_count, g, rem);
    const uint32_t left = (uint32_t)tn_count - pw * g;
    const uint32_t gs = left < pw ? left : pw;
    fast_divmod(rem, gs, q, r);
    tm = q;
    tn = pw * g + r;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Diagnostic build only (-DKR_STAMP, tools/stamp_build.sh): per-wave cycle sums of the three parts of an output tile in gemm_nt_pingpong — [0] tile
// set-up (coordinates, cursor), [1] the K loop (all intervals), [2] the epilogue, [3] tiles — accumulated into kr_stamp_buf[wave 0..7][4].  The
// product build contains none of this.
#ifdef KR_STAMP
__device__ unsigned long long kr_stamp_buf[8 * 8 * 4];   // [slot][wave][part]
__device__ int kr_stamp_slot;                              // set by the host (stream-ordered) before a launch
__device__ unsigned long long kr_stamp_fine[8 * 2 * 8];    // [slot][group][step]: cycle sums at finer points inside an epilogue (kept in SGPRs per tile)
#define KR_STAMP_NOW() __builtin_amdgcn_s_memtime()
#endif

template <class T, class Shape, int STAGES = 3, bool SWAP = false, class Coord, class Epilogue>
__device__ __forceinline__ void gemm_nt_stream(const uint16_t* __restrict__ A, int64_t lda, int64_t M, const uint16_t* __restrict__ B, int64_t ldb,
                                               int64_t N, int K, int64_t total_tiles, char* smem, Coord&& coord, Epilogue&& epi) {
    constexpr int BM = Shape::BM, BN = Shape::BN, BK = Shape::BK;
    static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
    constexpr int GL = Shape::A_PIECES + Shape::B_PIECES;   // LDS-DMA instructions per wave per K-tile
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / Shape::WN, wn = wave % Shape::WN;
    const int64_t G = gridDim.x;
    const int64_t my = (total_tiles > (int64_t)blockIdx.x) ? (total_tiles - blockIdx.x + G - 1) / G : 0;
    if (my == 0) return;
    const int nk = K / BK;
    const int64_t total_g = my * nk;

    // ---- prefetch cursor ---------------------------------------------------------------------------------------
    const char* pfA; const char* pfB;
    uint32_t a_off[Shape::A_PIECES], b_off[Shape::B_PIECES];
    auto setup_src = [&](int64_t i) {
        int64_t m0, n0;
        coord(xcd_chunk_map((int64_t)blockIdx.x + i * G, total_tiles), m0, n0);
        pfA = reinterpret_cast<const char*>(A + m0 * lda);
        pfB = reinterpret_cast<const char*>(B + n0 * ldb);
        const int64_t a_left = M - m0, b_left = N - n0;
#pragma unroll
        for (int p = 0; p < Shape::A_PIECES; ++p) {
            int row = (wave + p * Shape::NWAVE) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            if (row >= a_left) row = (int)a_left - 1;
            a_off[p] = (uint32_t)(row * lda * 2 + chunk * 16);
        }
#pragma unroll
        for (int p = 0; p < Shape::B_PIECES; ++p) {
            int row = (wave + p * Shape::NWAVE) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            if (row >= b_left) row = (int)b_left - 1;
            b_off[p] = (uint32_t)(row * ldb * 2 + chunk * 16);
        }
    };
    int64_t pf_g = 0, pf_tile = 0;
    int pf_kt = 0, pf_slot = 0, cp_slot = 0, inflight = 0;   // ring slots of the next DMA / the next tile to multiply; staged-not-consumed tiles
    setup_src(0);
    auto stage_next = [&]() {
        char* sa = smem + pf_slot * Shape::STAGE_BYTES;
        char* sb = sa + Shape::A_BYTES;
        const int64_t kbyte = (int64_t)pf_kt * BK * 2;
#pragma unroll
        for (int p = 0; p < Shape::A_PIECES; ++p)
            __builtin_amdgcn_global_load_lds((gbl_void*)(pfA + kbyte + a_off[p]), (lds_void*)(sa + (wave + p * Shape::NWAVE) * 1024), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < Shape::B_PIECES; ++p)
            __builtin_amdgcn_global_load_lds((gbl_void*)(pfB + kbyte + b_off[p]), (lds_void*)(sb + (wave + p * Shape::NWAVE) * 1024), 16, 0, 0);
        ++pf_g; ++inflight;
        pf_slot = (pf_slot + 1 == STAGES) ? 0 : pf_slot + 1;
        if (++pf_kt == nk) { pf_kt = 0; if (++pf_tile < my) setup_src(pf_tile); }
    };

    const int frow = lane & 31, fh = lane >> 5;
    const int fswz = (frow >> 1) & 7;
    const int a_row_byte = (wm * (BM / Shape::WM) + frow) * 128;
    const int b_row_byte = (wn * (BN / Shape::WN) + frow) * 128;

#pragma unroll
    for (int p = 0; p < STAGES - 1; ++p)
        if (pf_g < total_g) stage_next();

    for (int64_t i = 0; i < my; ++i) {
        int64_t m0, n0;
        const int64_t nat = xcd_chunk_map((int64_t)blockIdx.x + i * G, total_tiles);
        coord(nat, m0, n0);
        AccTile<Shape> acc;
        acc.m_wave = wm * (BM / Shape::WM);
        acc.n_wave = wn * (BN / Shape::WN);
        acc.lane = lane;
#pragma unroll
        for (int mi = 0; mi < Shape::TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < Shape::TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.v[mi][ni][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt) {
            // retire the oldest staged tile, leave the younger ones (at most STAGES-2) in flight
            if (STAGES >= 4 && inflight >= 3) wait_vmcnt<2 * GL>();
            else if (STAGES >= 3 && inflight >= 2) wait_vmcnt<GL>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            --inflight;   // the tile multiplied below is retired from the DMA queue (its buffer is re-staged only after the NEXT barrier)
            if (pf_g < total_g) stage_next();
            const char* sa = smem + cp_slot * Shape::STAGE_BYTES;
            cp_slot = (cp_slot + 1 == STAGES) ? 0 : cp_slot + 1;
            const char* sb = sa + Shape::A_BYTES;
            uint4 af[2][Shape::TM], bf[2][Shape::TN];
            auto load_frags = [&](int ks, int slot) {
                const int coff = ((2 * ks + fh) ^ fswz) << 4;
#pragma unroll
                for (int mi = 0; mi < Shape::TM; ++mi) af[slot][mi] = *reinterpret_cast<const uint4*>(sa + a_row_byte + mi * 32 * 128 + coff);
#pragma unroll
                for (int ni = 0; ni < Shape::TN; ++ni) bf[slot][ni] = *reinterpret_cast<const uint4*>(sb + b_row_byte + ni * 32 * 128 + coff);
            };
            load_frags(0, 0);
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                if (ks + 1 < BK / 16) load_frags(ks + 1, (ks + 1) & 1);
#pragma unroll
                for (int mi = 0; mi < Shape::TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < Shape::TN; ++ni)
                        acc.v[mi][ni] = SWAP ? T::mfma(bf[ks & 1][ni], af[ks & 1][mi], acc.v[mi][ni]) : T::mfma(af[ks & 1][mi], bf[ks & 1][ni], acc.v[mi][ni]);
            }
        }
        epi(acc, m0, n0, nat);
    }
}

// =====================================================================================================================
// split main loop: PRODUCER / CONSUMER waves on a 128x128x64 tile, for launches whose 256x256 tiling has too few tiles to fill the chip.
// A block has 8 waves (two per SIMD): waves 0..3 only multiply (64x64 each: per K-tile 16 MFMAs + the 16 ds_read_b128 of the fragments two
// k-steps ahead, nothing else), waves 4..7 only stage (8 LDS-DMA pieces each per K-tile, counted waits).  A wave issues in order, so in the
// streaming loop above every DMA issue (~45 cycles per 1-KiB piece), every LDS round trip at a K-tile start and every barrier skew is time in
// which that wave's SIMD issues no MFMA; with one block per CU nothing covers it (the 128x128 streaming loop reaches ~50 % MFMA duty).  Here
// the matrix pipe of each SIMD has a wave that never touches global memory.
// Ring of 4 K-tiles (4 x 32 KiB), ONE s_barrier per K-tile g (plus one start barrier), executed by all 8 waves:
//   producers: wait until tiles g and g+1 have landed (vmcnt: only tile g+2 may be in flight) -> barrier g -> issue tile g+3 into slot (g+3)&3
//              = the slot of tile g-1, which every consumer has finished reading before it arrived at barrier g;
//   consumers: barrier g -> k-steps 0..3 of tile g; the fragments of k-step s+2 are loaded during k-step s, so during k-steps 2, 3 they come
//              from tile g+1, which barrier g already guarantees to be resident: no LDS latency is exposed at a K-tile boundary.
// The K-tile stream runs across output tiles; producers run up to 3 K-tiles ahead while the consumers are in an epilogue.
// Accumulation order per output element is the same as in the other two loops (K-tiles in order, k-steps in order): results are bit-identical.
// =====================================================================================================================
using ShapeSplit = GemmShape<128, 128, 2, 2>;   // the consumer waves' view of the tile (the block has 2 x NWAVE waves)
constexpr int SPLIT_RING = 4;
constexpr int SPLIT_THREADS = 512;

template <class T, bool SWAP = false, class Coord, class Epilogue>
__device__ __forceinline__ void gemm_nt_split(const uint16_t* __restrict__ A, int64_t lda, int64_t M, const uint16_t* __restrict__ B, int64_t ldb,
                                              int64_t N, int K, int64_t total_tiles, char* smem, Coord&& coord, Epilogue&& epi) {
    using Shape = ShapeSplit;
    constexpr int BK = Shape::BK, STAGE = Shape::STAGE_BYTES, ABYTES = Shape::A_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t G = gridDim.x;
    const int64_t my = (total_tiles > (int64_t)blockIdx.x) ? (total_tiles - blockIdx.x + G - 1) / G : 0;
    if (my == 0) return;
    const int nk = K / BK;
    const int64_t total_g = my * nk;

    if (wave >= 4) {
        // ---------------------------------------------------------------- producers ----------------------------------------------------------------
        const int lw = wave - 4;                       // pieces lw, lw+4, lw+8, lw+12 of A and of B (a piece = 8 rows x 128 B = 1 KiB)
        const char* pfA; const char* pfB;
        uint32_t a_off[4], b_off[4];
        auto setup_src = [&](int64_t i) {
            int64_t m0, n0;
            coord(xcd_chunk_map((int64_t)blockIdx.x + i * G, total_tiles), m0, n0);
            pfA = reinterpret_cast<const char*>(A + m0 * lda);
            pfB = reinterpret_cast<const char*>(B + n0 * ldb);
            const int64_t a_left = M - m0, b_left = N - n0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int row = (lw + 4 * p) * 8 + (lane >> 3);
                const int chunk = (lane & 7) ^ ((row >> 1) & 7);
                const int ra = row >= a_left ? (int)a_left - 1 : row, rb = row >= b_left ? (int)b_left - 1 : row;
                a_off[p] = (uint32_t)(ra * lda * 2 + chunk * 16);
                b_off[p] = (uint32_t)(rb * ldb * 2 + chunk * 16);
            }
        };
        int64_t pf_g = 0, pf_tile = 0;
        int pf_kt = 0;
        setup_src(0);
        auto stage_next = [&]() {
            char* sa = smem + (int)(pf_g & (SPLIT_RING - 1)) * STAGE;
            char* sb = sa + ABYTES;
            const int64_t kbyte = (int64_t)pf_kt * BK * 2;
#pragma unroll
            for (int p = 0; p < 4; ++p)
                __builtin_amdgcn_global_load_lds((gbl_void*)(pfA + kbyte + a_off[p]), (lds_void*)(sa + (lw + 4 * p) * 1024), 16, 0, 0);
#pragma unroll
            for (int p = 0; p < 4; ++p)
                __builtin_amdgcn_global_load_lds((gbl_void*)(pfB + kbyte + b_off[p]), (lds_void*)(sb + (lw + 4 * p) * 1024), 16, 0, 0);
            ++pf_g;
            if (++pf_kt == nk) { pf_kt = 0; if (++pf_tile < my) setup_src(pf_tile); }
        };
#pragma unroll
        for (int p = 0; p < SPLIT_RING - 1; ++p)
            if (pf_g < total_g) stage_next();
        if (total_g > 2) wait_vmcnt<8>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                                        // start barrier: tiles 0 and 1 are resident (the consumers load their first fragments)
        for (int64_t g = 0; g < total_g; ++g) {
            if (g + 2 < total_g) wait_vmcnt<8>(); else wait_vmcnt<0>();      // tiles g and g+1 landed; g+2 may still be in flight
            __builtin_amdgcn_s_barrier();
            if (pf_g < total_g) stage_next();                                // tile g+3 -> the slot of tile g-1
        }
        return;
    }

    // -------------------------------------------------------------------- consumers ----------------------------------------------------------------
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 31, fh = lane >> 5;
    const int fswz = (frow >> 1) & 7;
    const int a_row_byte = (wm * 64 + frow) * 128;
    const int b_row_byte = ABYTES + (wn * 64 + frow) * 128;
    uint4 af[4][2], bf[4][2];                              // fragment buffers, one per k-step of a K-tile (loaded two k-steps ahead)
    auto load_frags = [&](int64_t g, int ks, int buf) {
        const char* st = smem + (int)(g & (SPLIT_RING - 1)) * STAGE;
        const int coff = ((2 * ks + fh) ^ fswz) << 4;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[buf][mi] = *reinterpret_cast<const uint4*>(st + a_row_byte + mi * 32 * 128 + coff);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bf[buf][ni] = *reinterpret_cast<const uint4*>(st + b_row_byte + ni * 32 * 128 + coff);
    };
    int64_t g = 0;
    __builtin_amdgcn_s_barrier();                          // start barrier (see the producers)
    load_frags(0, 0, 0); load_frags(0, 1, 1);
    for (int64_t i = 0; i < my; ++i) {
        int64_t m0, n0;
        const int64_t nat = xcd_chunk_map((int64_t)blockIdx.x + i * G, total_tiles);
        coord(nat, m0, n0);
        AccTile<Shape> acc;
        acc.m_wave = wm * 64;
        acc.n_wave = wn * 64;
        acc.lane = lane;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc.v[mi][ni][r] = 0.f;
        for (int kt = 0; kt < nk; ++kt, ++g) {
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                // unconditional (no branch in the loop body, so the compiler keeps COUNTED lgkmcnt waits): past the end of the stream the reads
                // hit a stale ring slot and are never used
                if (ks < 2) load_frags(g, ks + 2, ks + 2);
                else load_frags(g + 1, ks - 2, ks - 2);
                __builtin_amdgcn_sched_barrier(0);         // keep the two-k-step prefetch distance (the scheduler would sink the reads next to their use)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc.v[mi][ni] = SWAP ? T::mfma(bf[ks][ni], af[ks][mi], acc.v[mi][ni]) : T::mfma(af[ks][mi], bf[ks][ni], acc.v[mi][ni]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        epi(acc, m0, n0, nat);
    }
}

// =====================================================================================================================
// skinny main loop: ONE (32 WM) x (32 WN) output tile per block, for launches with a handful of token rows (a query or two, the reference's batches of
// 4-8 passages, the KiRAG loop's chain queries).  Such a GEMM is a latency chain, not a throughput problem: a CU that streams B bytes of operands with R
// bytes in flight needs (B / R) memory round trips, so the time falls with the number of CUs the operand stream is spread over — 32 x 32 tiles give
// F/32 x T/32 of them (FF2 at 32 tokens: 32 CUs x 512 KiB instead of 8 CUs x 2 MiB with 128x128 tiles) — and with the ring depth.  Split-K would spread
// further but changes the summation order; here every output element is still ONE accumulator chain over k in increasing order with the same MFMA
// instruction, so rows stay bit-identical to the other loops.
// Block = WM*WN multiplying waves (each owns one 32x32 sub-tile: 2 ds_read_b128 + 1 MFMA per k-step, fragments two k-steps ahead) + 4 staging waves
// (per K-tile 4 WM + 4 WN pieces of 1 KiB: 32 rows x 128 B per 32 rows of each operand) into a ring of RING K-tiles; one s_barrier per K-tile, same
// protocol as gemm_nt_split with the ring depth as the only difference (tiles g, g+1 resident at barrier g, tile g+RING-1 issued after it).
//   1 x 1 (32 x 32):  launches with at most one such tile per CU — RING 16 x 8 KiB = 128 KiB, 13 K-tiles in flight, one block per CU
//   2 x 2 (64 x 64):  more tiles than CUs (round 5; before: 32 x 32 tiles with a 4-deep ring, up to five blocks per CU): the four multiplying waves share
//                     every staged K-tile, i.e. HALF the L2 -> LDS bytes per output element — FF2 at 1024 tokens moved 512 MiB through 32 x 32 tiles
//                     (1024 tiles x 512 KiB) and was bound by exactly that; RING 8 x 16 KiB = 128 KiB
// Round 5 (profiles/r05/tried_skinny_prologue.txt): the epilogue's constants are requested by `pre` before the first barrier; issuing weight pieces BEFORE
// the wait for the row count (*Mp lives in device memory) was built and measured: the first two K-tiles' +-0, all prologue tiles' +2 us per launch (vmcnt
// retires in order: the first A piece then stands behind 15 cold weight pieces) — the order stays (A, B) tile by tile behind the row count.
// =====================================================================================================================
using ShapeSkinny = GemmShape<32, 32, 1, 1>;
constexpr int SKINNY_STAGE = 8192;                         // bytes per K-tile of the 1 x 1 shape
#ifndef SKINNY_GROUP
#define SKINNY_GROUP 2                                     // K-tiles per barrier at most (4 needs a ring of 16 and measured the same as 2: profiles/r05/tried_skinny_prologue.txt); 1 = rounds 1-4
#endif
template <int WM, int WN> struct SkinnyGeom {
    static constexpr int NC = WM * WN;                     // multiplying waves
    static constexpr int THREADS = (NC + 4) * 64;
    static constexpr int A_BYTES = WM * 4096, B_BYTES = WN * 4096, STAGE = A_BYTES + B_BYTES;
    static constexpr int PPW = WM + WN;                    // 1-KiB pieces per staging wave and K-tile
};
constexpr int SKINNY_THREADS = SkinnyGeom<1, 1>::THREADS;

// s_waitcnt vmcnt(n) for a run-time n (the instruction takes an immediate): FULL — the steady-state count of the ring, ONE compare on the path that runs
// once per K-tile (a 15-step compare ladder there cost the staging waves more cycles per K-tile than the multiplying wave's four MFMAs: +1.3 us per
// launch at 32 tokens) — or, in the ring's fill and drain, the largest listed value <= n
template <int FULL>
__device__ __forceinline__ void wait_vmcnt_upto(int n) {
    if (n >= FULL) { wait_vmcnt<FULL>(); return; }
    if (n >= 16) { if (n >= 24) wait_vmcnt<24>(); else if (n >= 20) wait_vmcnt<20>(); else wait_vmcnt<16>(); }
    else if (n >= 8) { if (n >= 12) wait_vmcnt<12>(); else if (n >= 10) wait_vmcnt<10>(); else wait_vmcnt<8>(); }
    else if (n >= 4) { if (n >= 6) wait_vmcnt<6>(); else wait_vmcnt<4>(); }
    else if (n >= 2) { if (n >= 3) wait_vmcnt<3>(); else wait_vmcnt<2>(); }
    else if (n >= 1) wait_vmcnt<1>(); else wait_vmcnt<0>();
}

// returns false (every wave, before any barrier) when the block's rows lie beyond *Mp
template <class T, int RING, int WM = 1, int WN = 1, bool SWAP = false, class Pre, class Epilogue>
__device__ __forceinline__ bool gemm_nt_skinny(const uint16_t* __restrict__ A, int64_t lda, const int* __restrict__ Mp, int64_t m0, const uint16_t* __restrict__ B, int64_t ldb,
                                               int64_t N, int64_t n0, int K, char* smem, Pre&& pre, Epilogue&& epi) {
    using G = SkinnyGeom<WM, WN>;
    static_assert(RING >= 8 && (RING & (RING - 1)) == 0 && G::PPW * (RING - 1) <= 60, "ring depth: a power of two, its pieces countable by s_waitcnt vmcnt");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = K / 64;
    if (wave >= G::NC) {
        // ---------------------------------------------------------------- producers ----------------------------------------------------------------
        const int lw = wave - G::NC;                   // pieces lw, lw + 4, ... of A and of B (a piece = 8 rows x 128 B)
        const int r8 = lane >> 3;
        const int64_t b_left = N - n0;
        const char* pb[WN];
#pragma unroll
        for (int i = 0; i < WN; ++i) {
            const int row = (lw + 4 * i) * 8 + r8;
            const int64_t rb = row >= b_left ? b_left - 1 : row;
            pb[i] = reinterpret_cast<const char*>(B + (n0 + rb) * ldb) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
        const int64_t M = *Mp;
        if (m0 >= M) return false;
        const int64_t a_left = M - m0;
        const char* pa[WM];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int row = (lw + 4 * i) * 8 + r8;
            const int64_t ra = row >= a_left ? a_left - 1 : row;
            pa[i] = reinterpret_cast<const char*>(A + (m0 + ra) * lda) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
        int pf = 0;
        auto stage_next = [&]() {
            char* st = smem + (pf & (RING - 1)) * G::STAGE;
#pragma unroll
            for (int i = 0; i < WM; ++i) __builtin_amdgcn_global_load_lds((gbl_void*)(pa[i] + (int64_t)pf * 128), (lds_void*)(st + (lw + 4 * i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < WN; ++i) __builtin_amdgcn_global_load_lds((gbl_void*)(pb[i] + (int64_t)pf * 128), (lds_void*)(st + G::A_BYTES + (lw + 4 * i) * 1024), 16, 0, 0);
            ++pf;
        };
        // pieces that may stay in flight when K-tiles 0 .. need must have landed
        auto allowed = [&](int need) { return G::PPW * (pf - 1 - need); };
        // GRP K-tiles per barrier (round 5): the multiplying waves' four MFMAs per K-tile are 128 cycles, a barrier round trip with five to eight waves is
        // half of that again.  At barrier j tiles GRP j .. GRP j + GRP - 1 (multiplied in full) and GRP j + GRP (its first fragments are prefetched) are
        // resident; behind it the slots of the previous GRP tiles are free and GRP new tiles are issued.  Same MFMAs in the same order.
        auto grouped = [&](auto grp_tag) {
            constexpr int GRP = decltype(grp_tag)::value;
            for (int p = 0; p < RING - GRP; ++p)
                if (pf < nk) stage_next();
            constexpr int FULLG = G::PPW * (RING - 2 * GRP - 1);   // tiles GRP j .. GRP j + GRP landed, the younger ones in flight
            wait_vmcnt_upto<FULLG>(allowed(nk > GRP ? GRP : nk - 1));
            __builtin_amdgcn_s_barrier();                                    // start barrier
            for (int g = 0; g < nk; g += GRP) {
                wait_vmcnt_upto<FULLG>(allowed(g + GRP < nk ? g + GRP : nk - 1));
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < GRP; ++i)
                    if (pf < nk) stage_next();                               // -> the slots of tiles g - GRP .. g - 1
            }
        };
        if (SKINNY_GROUP >= 4 && RING >= 16 && (nk & 3) == 0) { grouped(std::integral_constant<int, 4>{}); return true; }
        if (SKINNY_GROUP >= 2 && (nk & 1) == 0) { grouped(std::integral_constant<int, 2>{}); return true; }
        for (int p = 0; p < RING - 1; ++p)
            if (pf < nk) stage_next();                 // tile by tile (A, B): a tile is complete as soon as its own pieces have landed (vmcnt retires in order)
        constexpr int FULL = G::PPW * (RING - 3);      // tiles g, g+1 landed, RING - 3 younger ones in flight
        wait_vmcnt_upto<FULL>(allowed(nk > 1 ? 1 : 0));
        __builtin_amdgcn_s_barrier();                                        // start barrier: tiles 0 and 1 resident
        for (int g = 0; g < nk; ++g) {
            wait_vmcnt_upto<FULL>(allowed(g + 1 < nk ? g + 1 : nk - 1));     // tiles g, g+1 landed; younger ones may be in flight
            __builtin_amdgcn_s_barrier();
            if (pf < nk) stage_next();                                       // tile g + RING - 1 -> the slot of tile g - 1
        }
        return true;
    }
    // -------------------------------------------------------------------- consumers ----------------------------------------------------------------
    const int wm = wave / WN, wn = wave % WN;
    pre(m0 + wm * 32, n0 + wn * 32);                       // the epilogue's constants are requested now and land under the main loop
    if (m0 >= (int64_t)*Mp) return false;
    const int frow = lane & 31, fh = lane >> 5;
    const int fswz = (frow >> 1) & 7;
    const int a_byte = (wm * 32 + frow) * 128, b_byte = G::A_BYTES + (wn * 32 + frow) * 128;
    uint4 af[4], bf[4];
    auto load_frags = [&](int g, int ks, int buf) {
        const char* st = smem + (g & (RING - 1)) * G::STAGE + (((2 * ks + fh) ^ fswz) << 4);
        af[buf] = *reinterpret_cast<const uint4*>(st + a_byte);
        bf[buf] = *reinterpret_cast<const uint4*>(st + b_byte);
    };
    AccTile<ShapeSkinny> acc;
    acc.m_wave = 0; acc.n_wave = 0; acc.lane = lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc.v[0][0][r] = 0.f;
    __builtin_amdgcn_s_barrier();                          // start barrier
    load_frags(0, 0, 0); load_frags(0, 1, 1);
    auto grouped = [&](auto grp_tag) {
        constexpr int GRP = decltype(grp_tag)::value;
        for (int g = 0; g < nk; g += GRP) {
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int ks = 0; ks < 4 * GRP; ++ks) {         // k-steps of tiles g .. g + GRP - 1; the fragments of k-step ks + 2 are requested now
                load_frags(g + ((ks + 2) >> 2), (ks + 2) & 3, (ks + 2) & 3);      // the last two: tile g + GRP (past the end: a stale slot, never used)
                __builtin_amdgcn_sched_barrier(0);
                acc.v[0][0] = SWAP ? T::mfma(bf[ks & 3], af[ks & 3], acc.v[0][0]) : T::mfma(af[ks & 3], bf[ks & 3], acc.v[0][0]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        epi(acc, m0 + wm * 32, n0 + wn * 32);
    };
    if (SKINNY_GROUP >= 4 && RING >= 16 && (nk & 3) == 0) { grouped(std::integral_constant<int, 4>{}); return true; }
    if (SKINNY_GROUP >= 2 && (nk & 1) == 0) { grouped(std::integral_constant<int, 2>{}); return true; }
    for (int g = 0; g < nk; ++g) {
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 2) load_frags(g, ks + 2, ks + 2);
            else load_frags(g + 1, ks - 2, ks - 2);        // past the end: a stale slot, never used
            __builtin_amdgcn_sched_barrier(0);
            acc.v[0][0] = SWAP ? T::mfma(bf[ks], af[ks], acc.v[0][0]) : T::mfma(af[ks], bf[ks], acc.v[0][0]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    epi(acc, m0 + wm * 32, n0 + wn * 32);
    return true;
}

// =====================================================================================================================
// v3 main loop: PING-PONG.  256x256x64 tiles, 8 waves = two groups of four (group = wave >> 2 owns tile rows
// [128*group, +128), wave & 3 owns 64 columns; waves w and w+4 share a SIMD, so every SIMD hosts one wave of each group).
// Time is cut into intervals separated by ONE s_barrier each; in every interval one group multiplies (16 MFMAs of
// 32x32x16 = 512 cycles of its SIMD's matrix pipe, nothing else) while the other group does everything else: the 12
// ds_read_b128 of its next 16 MFMAs, 4 LDS-DMA pieces of a later K-tile, and the counted waits.  The matrix pipe of every
// SIMD therefore always has a wave whose only job is to feed it, and no LDS / DMA latency sits between two MFMAs.
//
//   interval        4t-1      4t        4t+1      4t+2      4t+3 ...
//   group 0         L(t,0)    M(t,0)    L(t,1)    M(t,1)    L(t+1,0)
//   group 1         M(t-1,1)  L(t,0)    M(t,0)    L(t,1)    M(t,1)
//
// L(t,h) reads the fragments of k-half h of K-tile t (ring slot t & 1) and issues DMA:  group 0: L(t,0) -> B rows 0..127 of
// tile t+1, L(t,1) -> A rows 128..255 of tile t+1;  group 1: L(t,0) -> B rows 128..255 of tile t+1, L(t,1) -> A rows 0..127
// of tile t+2.  Each L ends with  s_waitcnt vmcnt(4)  (everything but the 4 pieces just issued has landed: every piece gets
// >= 2 intervals of latency budget and lands one barrier before its first reader, cdna_hip_programming.md "Read a staged
// buffer one phase AFTER the wait that retires it") and  lgkmcnt(0)  (this wave's fragment reads are done, so the region it
// read may be re-staged after the barrier: WAR).  Region life times (K-tile t in slot t & 1):
//   A rows 0..127   read by group 0 in 4t-1, 4t+1     re-staged (tile t+2) by group 1 in 4t+2
//   A rows 128..255 read by group 1 in 4t,   4t+2     re-staged by group 0 in 4t+5
//   B               read by both in 4t-1 .. 4t+2      re-staged by group 0 in 4t+3 (rows 0..127), group 1 in 4t+4
// SWAP = true multiplies B.A^T instead (operands exchanged in the MFMA): accumulator tile (mi, ni) then holds the TRANSPOSED 32x32 block,
// element (register r, lane l) = C[m_wave + mi*32 + (l & 31)][n_wave + ni*32 + (r & 3) + 8 (r >> 2) + 4 (l >> 5)]: a lane owns 4 consecutive
// columns of one row, which is what a row-major 16-bit epilogue wants (packed 8-byte LDS-stage writes).
// The K-tile stream runs across output tiles (persistent blocks, as v2); past the end of the stream the DMA cursors stay
// on the last K-tile (dummy re-loads into regions nobody reads any more) so that the vmcnt arithmetic never changes.
// =====================================================================================================================
using ShapePP = GemmShape<256, 256, 2, 4>;

// M16 (experiment, tools/gemm_bench.hip only): the same loop on v_mfma_f32_16x16x32 — one 32-deep k-step per interval, 8 x 4 blocks of 16 x 16 per wave,
// the same 12 ds_read_b128 and the same matrix-pipe cycles per interval (MI355X_MICROARCH.md, DVFS give-back item 7: the chip may hold a higher clock on
// this shape).  The accumulators are handed to the epilogue in AccTile's storage with block (mi16, ni16) in v[mi16 >> 1][ni16 >> 1] registers
// 8 (mi16 & 1) + 4 (ni16 & 1) .. + 3 — a layout no product epilogue understands.
template <class T, bool SWAP = false, bool A_NT = false, bool M16 = false, class Coord, class Epilogue>
__device__ __forceinline__ void gemm_nt_pingpong(const uint16_t* __restrict__ A, int64_t lda, int64_t M, const uint16_t* __restrict__ B, int64_t ldb,
                                                 int64_t N, int K, int64_t total_tiles, char* smem, Coord&& coord, Epilogue&& epi) {
    using Shape = ShapePP;
    constexpr int BK = Shape::BK, STAGE = Shape::STAGE_BYTES, ABYTES = Shape::A_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3;
    const int64_t G = gridDim.x;
    const int64_t my = (total_tiles > (int64_t)blockIdx.x) ? (total_tiles - blockIdx.x + G - 1) / G : 0;
    if (my == 0) return;
    const int nk = K / BK;
    const int64_t total_g = my * nk;

    // ---- DMA cursors: a position in this block's K-tile stream.  A piece's address = a buffer resource whose base is the first row of the cursor's
    // output tile (scalar registers, re-made once per output tile) + the K-tile's byte offset (the instruction's scalar offset) + a 32-bit per-lane
    // offset that never changes (row inside the tile and swizzled 16-B chunk): buffer_load_dwordx4 ... lds.  Rows past the end of the operand (a partial
    // last tile) are out of the resource's range and arrive as zeros (their results are never read: see the callers), so nothing is clamped per tile.
    // (global_load_lds with 64-bit per-lane addresses cost 16 VGPRs for the two cursors' pieces, 64-bit VALU adds per piece in every interval and a
    // per-lane offset rebuild at every output tile.)
    struct Cursor {
        int64_t tile; int kt; int slot;      // output tile index (in this block's sequence), K-tile inside it, ring slot of the stream index
        __amdgpu_buffer_rsrc_t rs;
    };
    auto lane_off = [&](int piece0, int64_t ld, uint32_t (&off)[4]) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = (piece0 + p) * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            off[p] = (uint32_t)(row * ld * 2 + chunk * 16);
        }
    };
    auto set_base = [&](Cursor& c, bool isA) {
        int64_t m0, n0;
        coord(xcd_chunk_map((int64_t)blockIdx.x + c.tile * G, total_tiles), m0, n0);
        const int64_t r0 = isA ? m0 : n0, ld = isA ? lda : ldb, left = (isA ? M : N) - r0;
        const uint64_t base = reinterpret_cast<uint64_t>((isA ? A : B) + r0 * ld);
        const uint64_t bytes = (uint64_t)left * (uint64_t)ld * 2u;
        // the tile coordinates come out of VALU arithmetic (fast_divmod): tell the compiler they are wave-uniform
        const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        const uint32_t num = __builtin_amdgcn_readfirstlane(bytes > 0xffffffffull ? 0xffffffffu : (uint32_t)bytes);
        c.rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)bhi << 32) | blo), 0, (int)num, 0x00020000);
    };
    auto init_cursor = [&](Cursor& c, int64_t s, bool isA) {
        c.slot = (int)(s & 1);
        if (s >= total_g) s = total_g - 1;
        c.tile = s / nk; c.kt = (int)(s - c.tile * nk);
        set_base(c, isA);
    };
    auto issue4 = [&](Cursor& c, bool isA, int piece0, const uint32_t (&off)[4]) {
        char* dst = smem + c.slot * STAGE + (isA ? 0 : ABYTES) + piece0 * 1024;
        const int koff = c.kt * (BK * 2);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            // A_NT: the A operand is a once-through stream (the corpus of the coarse scan): non-temporal, so that it does not push the small,
            // re-used B operand (the query block) out of L2
            if (A_NT && isA) __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rs, (lds_void*)(dst + p * 1024), 16, off[p], koff, 0, 2);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rs, (lds_void*)(dst + p * 1024), 16, off[p], koff, 0, 0);
        }
        c.slot ^= 1;
        if (c.kt + 1 < nk) { ++c.kt; }
        else if (c.tile + 1 < my) { c.kt = 0; ++c.tile; set_base(c, isA); }
        // else: end of the stream, stay on the last K-tile (dummy re-loads)
    };

    const int xpiece = grp * 16 + wq * 4;          // B pieces this wave stages in L(t,0)
    const int ypiece = (1 - grp) * 16 + wq * 4;    // A pieces this wave stages in L(t,1)

    // ---- prologue: K-tile 0 completely, A rows 0..127 of K-tile 1 ---------------------------------------------------------
    {
        Cursor c;
        uint32_t pa[4], pb[4];
        lane_off(wave * 4, lda, pa); lane_off(wave * 4, ldb, pb);
        init_cursor(c, 0, true);  { Cursor d = c; issue4(d, true, wave * 4, pa); }
        init_cursor(c, 0, false); { Cursor d = c; issue4(d, false, wave * 4, pb); }
        // A rows 0..127 of tile 1 = 16 pieces = 2 per wave; issue4 moves 4 pieces, so waves 0..3 take them (pieces 4*wave .. +3)
        if (wave < 4) { init_cursor(c, 1, true); issue4(c, true, wave * 4, pa); }
    }
    Cursor cx, cy;
    uint32_t offx[4], offy[4];                     // per-lane piece offsets of the two cursors: fixed for the whole kernel
    lane_off(xpiece, ldb, offx); lane_off(ypiece, lda, offy);
    init_cursor(cx, 1, false);
    init_cursor(cy, grp ? 2 : 1, true);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();   // group 1 runs one interval behind group 0

    const int frow = lane & 31, fh = lane >> 5;
    const int fswz = (frow >> 1) & 7;
    const int a_row_byte = (grp * 128 + frow) * 128;
    const int b_row_byte = (wq * 64 + frow) * 128;

    int cur = 0;   // ring slot of the K-tile being multiplied
#ifdef KR_STAMP
    unsigned long long st_gap = 0, st_loop = 0, st_epi = 0, st_t0 = KR_STAMP_NOW();
    unsigned long long st_fine[6] = {0, 0, 0, 0, 0, 0};
#endif
    for (int64_t i = 0; i < my; ++i) {
        int64_t m0, n0;
        const int64_t nat = xcd_chunk_map((int64_t)blockIdx.x + i * G, total_tiles);
        coord(nat, m0, n0);
#ifdef KR_STAMP
        const unsigned long long st_t1 = KR_STAMP_NOW();
        st_gap += st_t1 - st_t0;
#endif
        AccTile<Shape> acc;
        acc.m_wave = grp * 128;
        acc.n_wave = wq * 64;
        acc.lane = lane;
        f32x4 c16[M16 ? 8 : 1][M16 ? 4 : 1];
        // One interval = L(t,h) (fragments of k-half h, 4 DMA pieces, counted waits) + barrier + M(t,h) (16 MFMAs, nothing else).  The first interval of
        // an output tile is a separate instantiation (FIRST): its first k-step takes the inline constant 0 as the C operand, so the accumulators are never
        // zero-initialised (128 v_mov per wave and tile at the boundary, where nothing overlaps them).
        auto interval = [&](auto first_tag, const char* sa, const char* sb, int h, bool last) {
            constexpr bool FIRST = decltype(first_tag)::value;
            if constexpr (M16) {
                uint4 af16[8], bf16v[4];
                const int r16 = lane & 15;
                const int coff16 = ((4 * h + (lane >> 4)) ^ ((r16 >> 1) & 7)) << 4;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) bf16v[ni] = *reinterpret_cast<const uint4*>(sb + (wq * 64 + ni * 16 + r16) * 128 + coff16);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) af16[mi] = *reinterpret_cast<const uint4*>(sa + (grp * 128 + mi * 16 + r16) * 128 + coff16);
                if (h == 0) issue4(cx, false, xpiece, offx); else issue4(cy, true, ypiece, offy);
                wait_vmcnt<4>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                        if (FIRST) c16[mi][ni] = SWAP ? T::mfma16(bf16v[ni], af16[mi], zero4) : T::mfma16(af16[mi], bf16v[ni], zero4);
                        else c16[mi][ni] = SWAP ? T::mfma16(bf16v[ni], af16[mi], c16[mi][ni]) : T::mfma16(af16[mi], bf16v[ni], c16[mi][ni]);
                    }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(last && grp)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                return;
            }
            uint4 af[2][4], bf[2][2];
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int coff = ((2 * (2 * h + k2) + fh) ^ fswz) << 4;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) bf[k2][ni] = *reinterpret_cast<const uint4*>(sb + b_row_byte + ni * 32 * 128 + coff);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) af[k2][mi] = *reinterpret_cast<const uint4*>(sa + a_row_byte + mi * 32 * 128 + coff);
            }
            if (h == 0) issue4(cx, false, xpiece, offx); else issue4(cy, true, ypiece, offy);
            wait_vmcnt<4>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        if constexpr (FIRST) {
                            if (k2 == 0) {
                                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                acc.v[mi][ni] = SWAP ? T::mfma(bf[k2][ni], af[k2][mi], zero) : T::mfma(af[k2][mi], bf[k2][ni], zero);
                                continue;
                            }
                        }
                        acc.v[mi][ni] = SWAP ? T::mfma(bf[k2][ni], af[k2][mi], acc.v[mi][ni]) : T::mfma(af[k2][mi], bf[k2][ni], acc.v[mi][ni]);
                    }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            // The barrier that ends an output tile's last M interval: group 0 passes it and then runs its epilogue, group 1
            // (whose M interval is the one AFTER it) runs its epilogue first and only then arrives, so the two epilogues run
            // side by side (VALU / LDS / store work of two waves per SIMD interleaves) instead of one after the other.
            if (!(last && grp)) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        {
            const char* sa = smem + cur * STAGE;
            cur ^= 1;
            interval(std::true_type{}, sa, sa + ABYTES, 0, false);
            interval(std::false_type{}, sa, sa + ABYTES, 1, nk == 1);
        }
        for (int kt = 1; kt < nk; ++kt) {
            const char* sa = smem + cur * STAGE;
            cur ^= 1;
            interval(std::false_type{}, sa, sa + ABYTES, 0, false);
            interval(std::false_type{}, sa, sa + ABYTES, 1, kt == nk - 1);
        }
#ifdef KR_STAMP
        const unsigned long long st_t2 = KR_STAMP_NOW();
        st_loop += st_t2 - st_t1;
        acc.fine_prev = st_t2;
#endif
        if constexpr (M16) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc.v[mi >> 1][ni >> 1][8 * (mi & 1) + 4 * (ni & 1) + e] = c16[mi][ni][e];
        }
        epi(acc, m0, n0, nat);
#ifdef KR_STAMP
        st_t0 = KR_STAMP_NOW();
        st_epi += st_t0 - st_t2;
        for (int z = 0; z < 6; ++z) st_fine[z] += acc.fine[z];
#endif
        if (grp) __builtin_amdgcn_s_barrier();
    }
#ifdef KR_STAMP
    if (lane == 0) {
        unsigned long long* sb = kr_stamp_buf + (kr_stamp_slot & 7) * 32 + wave * 4;
        atomicAdd(&sb[0], st_gap); atomicAdd(&sb[1], st_loop); atomicAdd(&sb[2], st_epi); atomicAdd(&sb[3], (unsigned long long)my);
        for (int z = 0; z < 6; ++z) atomicAdd(&kr_stamp_fine[((kr_stamp_slot & 7) * 2 + grp) * 8 + z], st_fine[z]);
    }
#endif
    if (!grp) __builtin_amdgcn_s_barrier();   // group 0 started one interval early: same number of barriers for every wave
    wait_vmcnt<0>();                            // dummy tail DMAs must not land after the caller re-uses the ring
    __builtin_amdgcn_s_barrier();
}


// =====================================================================================================================
// EXPERIMENT (tools/gemm_bench.hip only; not used by the product): ONE wave per SIMD.  256x256x64 tiles, 4 waves of 128x128 (256 accumulator registers
// of the 512 a lone wave may use), ring of two 64-KiB K-tiles, ONE s_barrier per K-tile: K-tile g+1 is staged (16 LDS-DMA pieces per wave: waves 0, 1
// the A rows, waves 2, 3 the B rows) while K-tile g is multiplied; the pieces are issued between the MFMAs of the first k-steps.  Purpose: measure what
// the main loop alone sustains in this structure before building the deferred-store epilogue that is its reason to exist.
// =====================================================================================================================
using ShapeSolo = GemmShape<256, 256, 2, 2>;

template <class T, bool SWAP = false, class Coord, class Epilogue>
__device__ __forceinline__ void gemm_nt_solo(const uint16_t* __restrict__ A, int64_t lda, int64_t M, const uint16_t* __restrict__ B, int64_t ldb,
                                             int64_t N, int K, int64_t total_tiles, char* smem, Coord&& coord, Epilogue&& epi) {
    using Shape = ShapeSolo;
    constexpr int BK = Shape::BK, STAGE = Shape::STAGE_BYTES, ABYTES = Shape::A_BYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t G = gridDim.x;
    const int64_t my = (total_tiles > (int64_t)blockIdx.x) ? (total_tiles - blockIdx.x + G - 1) / G : 0;
    if (my == 0) return;
    const int nk = K / BK;
    const int64_t total_g = my * nk;
    // ---- DMA cursor: this wave stages 16 pieces (128 rows) of ONE operand per K-tile: waves 0, 1 -> A rows 128 * wave .., waves 2, 3 -> B rows 128 * (wave - 2) ..
    const bool isA = wave < 2;
    const int row0 = (wave & 1) * 128;
    int64_t c_tile = 0; int c_kt = 0;
    const char* c_base = nullptr;
    uint32_t c_off[16];
    auto set_base = [&]() {
        int64_t m0, n0;
        coord(xcd_chunk_map((int64_t)blockIdx.x + c_tile * G, total_tiles), m0, n0);
        const int64_t r0 = isA ? m0 : n0, ld = isA ? lda : ldb, left = (isA ? M : N) - r0;
        c_base = reinterpret_cast<const char*>((isA ? A : B) + r0 * ld);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            int row = row0 + p * 8 + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            if (row >= left) row = (int)left - 1;
            c_off[p] = (uint32_t)(row * ld * 2 + chunk * 16);
        }
    };
    set_base();
    auto issue_pieces = [&](int slot, int p_begin, int p_end) {     // pieces [p_begin, p_end) of the cursor's K-tile into ring slot `slot`
        char* dst = smem + slot * STAGE + (isA ? 0 : ABYTES) + row0 * 128;
        const char* src = c_base + (int64_t)c_kt * (BK * 2);
#pragma unroll
        for (int p = p_begin; p < p_end; ++p)
            __builtin_amdgcn_global_load_lds((gbl_void*)(src + c_off[p]), (lds_void*)(dst + p * 1024), 16, 0, 0);
    };
    auto advance = [&]() {
        if (c_kt + 1 < nk) ++c_kt;
        else if (c_tile + 1 < my) { c_kt = 0; ++c_tile; set_base(); }
    };
    issue_pieces(0, 0, 16); advance();
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (total_g > 1) { issue_pieces(1, 0, 16); advance(); }            // K-tile 1 in flight while K-tile 0 is multiplied

    const int frow = lane & 31, fh = lane >> 5;
    const int fswz = (frow >> 1) & 7;
    const int a_row_byte = (wm * 128 + frow) * 128;
    const int b_row_byte = ABYTES + (wn * 128 + frow) * 128;
    int64_t g = 0;
    uint4 af[2][4], bf[2][4];
    auto load_frags = [&](int64_t gg, int ks, int buf) {
        const char* st = smem + (int)(gg & 1) * STAGE;
        const int coff = ((2 * ks + fh) ^ fswz) << 4;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[buf][mi] = *reinterpret_cast<const uint4*>(st + a_row_byte + mi * 32 * 128 + coff);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) bf[buf][ni] = *reinterpret_cast<const uint4*>(st + b_row_byte + ni * 32 * 128 + coff);
    };
    load_frags(0, 0, 0);
    for (int64_t i = 0; i < my; ++i) {
        int64_t m0, n0;
        const int64_t nat = xcd_chunk_map((int64_t)blockIdx.x + i * G, total_tiles);
        coord(nat, m0, n0);
        AccTile<Shape> acc;
        acc.m_wave = wm * 128; acc.n_wave = wn * 128; acc.lane = lane;
        // Software pipeline per K-tile g (fragments of its k-step 0 are already in buffer 0):
        //   k-steps 0, 1, 2   MFMAs + the fragment reads of the next k-step
        //   wait (own DMA pieces of K-tile g + 1 landed, own LDS reads done) + s_barrier: K-tile g + 1 is resident, nobody reads K-tile g any more
        //   issue the 16 DMA pieces of K-tile g + 2 into K-tile g's slot and read the fragments of (g + 1, k-step 0)  -- under the MFMAs of k-step 3
        // The first K-tile of an output tile is a separate instantiation (C operand 0 in its first k-step; a run-time branch there costs > 100 spills).
        auto mfmas = [&](auto first_tag, int ks) {
            constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    if constexpr (FIRST) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc.v[mi][ni] = SWAP ? T::mfma(bf[ks & 1][ni], af[ks & 1][mi], zero) : T::mfma(af[ks & 1][mi], bf[ks & 1][ni], zero);
                    } else {
                        acc.v[mi][ni] = SWAP ? T::mfma(bf[ks & 1][ni], af[ks & 1][mi], acc.v[mi][ni]) : T::mfma(af[ks & 1][mi], bf[ks & 1][ni], acc.v[mi][ni]);
                    }
                }
        };
        auto ktile = [&](auto first_tag) {
            load_frags(g, 1, 1);
            mfmas(first_tag, 0);
            load_frags(g, 2, 0);
            mfmas(std::false_type{}, 1);
            load_frags(g, 3, 1);
            mfmas(std::false_type{}, 2);
            wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (g + 2 < total_g) { issue_pieces((int)(g & 1), 0, 16); advance(); }
            if (g + 1 < total_g) load_frags(g + 1, 0, 0);
            mfmas(std::false_type{}, 3);
            ++g;
        };
        ktile(std::true_type{});
        for (int kt = 1; kt < nk; ++kt) ktile(std::false_type{});
        epi(acc, m0, n0, nat);
    }
}

}  // namespace kr
