/*
 * TEST ORACLE (see oracle/__init__.py) — plain-C restatement of the search half of the hot path.
 *
 * Reference call sites this follows:
 *   retriever/index.py:26-34   Indexer.index_data   -> append fp32 rows + int64 ids
 *   retriever/index.py:36-53   Indexer.search_knn   -> per 1024-query block: faiss IndexFlatIP.search(q, k),
 *                                                      scores descending, internal row -> external id
 * The arithmetic itself lives in faiss-cpu==1.8.0.post1 (requirements.txt:10), which is not vendored in the
 * reference and not installed here: PARITY WITH faiss IS UNPINNED.  This file restates IndexFlatIP's published
 * semantics (exact inner product over every stored row, k best per query, descending) and fixes what faiss
 * leaves to its BLAS: the rounding of the sum and the tie rule.
 *
 * Canonical score — a mathematical definition, independent of any summation order (and so of any kernel):
 *
 *      score(q, x) = RN32( sum_i q[i] * x[i] )         the EXACT real inner product of the fp32 inputs,
 *                                                      rounded ONCE to fp32, round-to-nearest-even
 *
 * (an fp32 BLAS sgemm — what faiss calls — returns this value perturbed by its own accumulation order, within
 * ~d * 2^-24 relative).  Two independent formulations are given here:
 *   kr_oracle_dot_exact   a fixed-point super-accumulator: every product (exact: 24 x 24 bits) is added as an
 *                         integer into a 704-bit accumulator, i = 0, 1, ..., d-1; the exact integer is rounded once.
 *   kr_oracle_dot         plain sequential fp64: S = fl(sum p_i) and A = fl(sum |p_i|), i = 0, 1, ..., d-1; the
 *                         textbook bound |S - exact| <= (d-1) u A / (1 - (d-1) u) (u = 2^-53; the products p_i are
 *                         exact in fp64) gives an interval [S - E, S + E] that contains the exact sum; when both
 *                         ends round to the same float that float IS the canonical score; otherwise (about 2 in
 *                         10^6 scores of unit vectors) the super-accumulator decides.
 * tests/test_oracle_search.py pins both against Python rationals (fractions.Fraction), incl. exact midpoints,
 * subnormals, cancellation and huge exponent spreads, and against the committed vectors tests/golden/g9_exact_dot.npz.
 *
 * Ranking: score descending, ties by internal row index ascending.  k > n is an error (-1).
 * Non-finite inputs: if sum |p_i| is not finite the score is (float)S (inf / NaN propagate as in IEEE arithmetic).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KR_NLIMB 22 /* 32 payload bits per int64 limb: bit 0 = 2^-298 (the smallest product of two fp32 subnormals) */

/* fp32 -> (sign, 24-bit integer mantissa m, exponent e) with |f| = m * 2^e ; finite inputs only */
static inline void decode_f32(float f, int* sign, uint64_t* m, int* e) {
    uint32_t u; memcpy(&u, &f, 4);
    *sign = (int)(u >> 31);
    const uint32_t ex = (u >> 23) & 0xffu, fr = u & 0x7fffffu;
    if (ex == 0) { *m = fr; *e = -149; }
    else { *m = fr | 0x800000u; *e = (int)ex - 150; }
}

/* RN32 of the exact inner product; inputs must be finite (callers check) */
float kr_oracle_dot_exact(const float* q, const float* x, int d) {
    int64_t acc[KR_NLIMB];
    memset(acc, 0, sizeof(acc));
    for (int i = 0; i < d; ++i) {
        int sa, sb, ea, eb; uint64_t ma, mb;
        decode_f32(q[i], &sa, &ma, &ea);
        decode_f32(x[i], &sb, &mb, &eb);
        const uint64_t M = ma * mb;                      /* < 2^48, exact */
        if (M == 0) continue;
        const int o = ea + eb + 298;                     /* bit offset of M's LSB: 0 .. 506 */
        const int limb = o >> 5, sh = o & 31;
        const unsigned __int128 v = (unsigned __int128)M << sh;   /* < 2^79 */
        const int64_t p0 = (int64_t)(uint64_t)(v & 0xffffffffu), p1 = (int64_t)(uint64_t)((v >> 32) & 0xffffffffu), p2 = (int64_t)(uint64_t)(v >> 64);
        if (sa ^ sb) { acc[limb] -= p0; acc[limb + 1] -= p1; acc[limb + 2] -= p2; }
        else { acc[limb] += p0; acc[limb + 1] += p1; acc[limb + 2] += p2; }
    }
    /* carry propagation (limbs may be negative): acc[i] in [0, 2^32) afterwards, the top limb carries the sign */
    int64_t w[KR_NLIMB];
    int negative = 0;
    for (int pass = 0; pass < 2; ++pass) {
        int64_t carry = 0;
        for (int i = 0; i < KR_NLIMB; ++i) {
            const int64_t t = (pass ? -acc[i] : acc[i]) + carry;
            carry = t >> 32;                             /* arithmetic shift = floor division */
            w[i] = t - carry * 4294967296LL;
        }
        if (carry >= 0) break;                           /* value >= 0 : done (carry out of the top limb is 0 for non-negative values) */
        negative = 1;                                    /* value < 0 : redo on the negated limbs */
    }
    int top = KR_NLIMB - 1;
    while (top >= 0 && w[top] == 0) --top;
    if (top < 0) return 0.0f;                            /* exact zero: +0 */
    int hb = 31; while (!((w[top] >> hb) & 1)) --hb;
    const int P = top * 32 + hb;                         /* index of the leading bit; value = mag * 2^-298 */
    int lsb = P - 23; if (lsb < 149) lsb = 149;          /* fp32 quantum: 24 significant bits, never below 2^-149 */
    /* mant = mag >> lsb (<= 24 bits), round bit, sticky */
    uint64_t mant = 0;
    for (int b = P; b >= lsb; --b) mant = (mant << 1) | (uint64_t)((w[b >> 5] >> (b & 31)) & 1);
    int rnd = 0, sticky = 0;
    if (lsb >= 1) {
        rnd = (int)((w[(lsb - 1) >> 5] >> ((lsb - 1) & 31)) & 1);
        for (int b = lsb - 2; b >= 0 && !sticky; --b) sticky |= (int)((w[b >> 5] >> (b & 31)) & 1);
    }
    if (rnd && (sticky || (mant & 1))) ++mant;
    const float r = ldexpf((float)mant, lsb - 298);      /* exact: mant <= 2^24, overflow -> inf, mant == 0 -> 0 */
    return negative ? -r : r;
}

/* canonical score by plain sequential fp64 with a certified rounding; falls back to the super-accumulator */
float kr_oracle_dot(const float* q, const float* x, int d) {
    double S = 0.0, A = 0.0;
    for (int i = 0; i < d; ++i) {
        const double p = (double)q[i] * (double)x[i];    /* exact */
        S += p; A += fabs(p);
    }
    if (!(A <= 1.7e308)) return (float)S;                /* inf / NaN in the inputs */
    const double E = (double)(d + 4) * 1.2e-16 * A;      /* >= (d-1) u A / (1 - (d-1) u) + the rounding of S -+ E ; u = 2^-53 = 1.11e-16 */
    const float lo = (float)(S - E), hi = (float)(S + E);
    uint32_t ul, uh; memcpy(&ul, &lo, 4); memcpy(&uh, &hi, 4);
    if (ul == uh) return lo;
    return kr_oracle_dot_exact(q, x, d);
}

/* the score an fp32 accumulation in index order gives (what a naive sgemm-free loop would return): used by the tests
 * to REPORT the distance of the canonical score from fp32 arithmetic (faiss's BLAS sits within the same bound). */
float kr_oracle_dot_f32(const float* q, const float* x, int d) {
    float s = 0.f;
    for (int i = 0; i < d; ++i) s += q[i] * x[i];
    return s;
}

/* a is "better" than b */
static inline int better(float sa, int64_t ia, float sb, int64_t ib) {
    return (sa > sb) || (sa == sb && ia < ib);
}

typedef struct { float s; int64_t i; } ent_t;

static void sift_down(ent_t* h, int n, int p) {
    for (;;) {
        int c = 2 * p + 1;
        if (c >= n) break;
        /* min-heap on "better": root = worst */
        if (c + 1 < n && better(h[c].s, h[c].i, h[c + 1].s, h[c + 1].i)) c = c + 1;
        if (better(h[p].s, h[p].i, h[c].s, h[c].i)) { ent_t t = h[p]; h[p] = h[c]; h[c] = t; p = c; }
        else break;
    }
}

static int cmp_desc(const void* a, const void* b) {
    const ent_t* x = (const ent_t*)a; const ent_t* y = (const ent_t*)b;
    if (better(x->s, x->i, y->s, y->i)) return -1;
    if (better(y->s, y->i, x->s, x->i)) return 1;
    return 0;
}

/* scores_out [nq,k] float, idx_out [nq,k] int64 (internal row numbers). returns 0, or -1 if k > n or k <= 0.
 * Rows whose score is NaN are never returned (an all-masked passage encodes to NaN, encoders.py:56-58); if fewer than
 * k rows have a real score the call fails with -2. */
int kr_oracle_search(const float* q, int nq, const float* x, int64_t n, int d, int k,
                     float* scores_out, int64_t* idx_out) {
    if (k <= 0 || (int64_t)k > n) return -1;
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int qi = 0; qi < nq; ++qi) {
        ent_t* h = (ent_t*)malloc(sizeof(ent_t) * (size_t)k);
        int hn = 0;
        const float* qv = q + (size_t)qi * d;
        for (int64_t r = 0; r < n; ++r) {
            float s = kr_oracle_dot(qv, x + (size_t)r * d, d);
            if (s != s) continue;
            if (hn < k) {
                h[hn].s = s; h[hn].i = r; ++hn;
                if (hn == k) for (int p = k / 2 - 1; p >= 0; --p) sift_down(h, k, p);
            } else if (better(s, r, h[0].s, h[0].i)) {
                h[0].s = s; h[0].i = r; sift_down(h, k, 0);
            }
        }
        if (hn < k) {
#pragma omp atomic write
            bad = 1;
        } else {
            qsort(h, (size_t)k, sizeof(ent_t), cmp_desc);
            for (int j = 0; j < k; ++j) { scores_out[(size_t)qi * k + j] = h[j].s; idx_out[(size_t)qi * k + j] = h[j].i; }
        }
        free(h);
    }
    return bad ? -2 : 0;
}

/* canonical scores of explicitly listed rows: out[nq, m] for rows[nq, m] (used to check re-rank inputs) */
void kr_oracle_scores_at(const float* q, int nq, const float* x, int d, const int64_t* rows, int m, float* out) {
#pragma omp parallel for schedule(static)
    for (int qi = 0; qi < nq; ++qi)
        for (int j = 0; j < m; ++j)
            out[(size_t)qi * m + j] = kr_oracle_dot(q + (size_t)qi * d, x + (size_t)rows[(size_t)qi * m + j] * d, d);
}

/* all-pairs scores out[nq, n] by the three formulations (which: 0 canonical, 1 exact super-accumulator, 2 fp32 loop) */
void kr_oracle_scores_all(const float* q, int nq, const float* x, int64_t n, int d, int which, float* out) {
#pragma omp parallel for schedule(static)
    for (int qi = 0; qi < nq; ++qi)
        for (int64_t r = 0; r < n; ++r) {
            const float* a = q + (size_t)qi * d; const float* b = x + (size_t)r * d;
            out[(size_t)qi * n + r] = which == 0 ? kr_oracle_dot(a, b, d) : which == 1 ? kr_oracle_dot_exact(a, b, d) : kr_oracle_dot_f32(a, b, d);
        }
}

/* round-to-nearest-even float -> bf16 bits (NaN kept NaN), the conversion the index applies to its coarse copy */
uint16_t kr_oracle_f32_to_bf16(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
