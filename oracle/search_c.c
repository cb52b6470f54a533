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
 * leaves to its BLAS: the summation order and the tie rule.
 *
 * Canonical score (shared bit-for-bit with the HIP re-rank kernel, kirag_amd/csrc/search.hip):
 *   acc[l] (double, l = 0..63) accumulates, in increasing i, (double)q[i]*(double)x[i] for the i with
 *   ((i >> 2) & 63) == l  (each product of two floats is exact in double);  then a 6-stage XOR butterfly
 *   p[l] = p[l] + p[l ^ m] for m = 32,16,8,4,2,1;  score = (float)p[0].
 * Ranking: score descending, ties by internal row index ascending.  k > n is an error (-1).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

double kr_oracle_dot(const float* q, const float* x, int d) {
    double acc[64];
    for (int l = 0; l < 64; ++l) acc[l] = 0.0;
    for (int i = 0; i < d; ++i) acc[(i >> 2) & 63] += (double)q[i] * (double)x[i];
    for (int m = 32; m >= 1; m >>= 1) {
        double nxt[64];
        for (int l = 0; l < 64; ++l) nxt[l] = acc[l] + acc[l ^ m];
        memcpy(acc, nxt, sizeof(acc));
    }
    return acc[0];
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

/* scores_out [nq,k] float, idx_out [nq,k] int64 (internal row numbers). returns 0, or -1 if k > n or k <= 0 */
int kr_oracle_search(const float* q, int nq, const float* x, int64_t n, int d, int k,
                     float* scores_out, int64_t* idx_out) {
    if (k <= 0 || (int64_t)k > n) return -1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int qi = 0; qi < nq; ++qi) {
        ent_t* h = (ent_t*)malloc(sizeof(ent_t) * (size_t)k);
        int hn = 0;
        const float* qv = q + (size_t)qi * d;
        for (int64_t r = 0; r < n; ++r) {
            float s = (float)kr_oracle_dot(qv, x + (size_t)r * d, d);
            if (hn < k) {
                h[hn].s = s; h[hn].i = r; ++hn;
                if (hn == k) for (int p = k / 2 - 1; p >= 0; --p) sift_down(h, k, p);
            } else if (better(s, r, h[0].s, h[0].i)) {
                h[0].s = s; h[0].i = r; sift_down(h, k, 0);
            }
        }
        qsort(h, (size_t)k, sizeof(ent_t), cmp_desc);
        for (int j = 0; j < k; ++j) { scores_out[(size_t)qi * k + j] = h[j].s; idx_out[(size_t)qi * k + j] = h[j].i; }
        free(h);
    }
    return 0;
}

/* canonical scores of explicitly listed rows: out[nq, m] for rows[nq, m] (used to check re-rank inputs) */
void kr_oracle_scores_at(const float* q, int nq, const float* x, int d, const int64_t* rows, int m, float* out) {
    for (int qi = 0; qi < nq; ++qi)
        for (int j = 0; j < m; ++j)
            out[(size_t)qi * m + j] = (float)kr_oracle_dot(q + (size_t)qi * d, x + (size_t)rows[(size_t)qi * m + j] * d, d);
}

/* round-to-nearest-even float -> bf16 bits (NaN kept NaN), the conversion the index applies to its coarse copy */
uint16_t kr_oracle_f32_to_bf16(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
