/* A C-only host that DRIVES THE GPU through include/kirag_amd.h — no Python, no torch, no HIP header: the INTEGRATION.md section B contract
 * (a maintainer binds libkirag_amd.so from any language with a C FFI; the reference's own call sites are retriever/index.py:19-24,32,47 and
 * retriever/encoders.py:67-77).  Everything crosses the boundary as HOST pointers, which the ABI accepts everywhere.
 *
 *   capi_gpu /path/to/libkirag_amd.so vectors.bin
 *
 * vectors.bin is written by tests/test_gpu_capi_c.py from the oracle (oracle/search_np.py canonical top-k, oracle/encoder_np.py e5_encode):
 *   "KRT1" | int32 n d nq k | float x[n*d] q[nq*d] | float exp_scores[nq*k] | int64 exp_rows[nq*k]
 *   | int32 hidden layers heads intermediate vocab max_pos type_vocab | float ln_eps
 *   | int32 ntensors | ntensors x { int32 name_len | name | int64 numel | float data[numel] }
 *   | int32 B S pool | int64 ids[B*S] mask[B*S] | float exp_emb[B*hidden] | float emb_tol
 * Index: create -> reserve -> add (two pieces) -> search (blocking) and search_async x 2 + finish_ex, rows / score bits compared with the oracle; get_rows.
 * Encoder: create_ex(f16, low half) and create_ex(bf16, none) -> load_weight x N -> finalize -> forward (host out) -> check, compared within emb_tol;
 * a token id outside the vocabulary must come back as KR_EINVAL. */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kirag_amd.h"

#define BIND(name)                                                    \
    __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); \
    if (!p_##name) { printf("missing symbol %s\n", #name); return 2; }
#define CHECK(call)                                                                                      \
    do { int rc_ = (call); if (rc_ != KR_OK) { printf("%s -> %d: %s\n", #call, rc_, p_kr_last_error()); return 1; } } while (0)

static int rd(FILE* f, void* p, size_t bytes) { return fread(p, 1, bytes, f) == bytes ? 0 : -1; }

int main(int argc, char** argv) {
    if (argc < 3) { printf("usage: capi_gpu /path/to/libkirag_amd.so vectors.bin\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 2; }
    BIND(kr_abi_version) BIND(kr_last_error) BIND(kr_device_count)
    BIND(kr_index_create) BIND(kr_index_destroy) BIND(kr_index_reserve) BIND(kr_index_add) BIND(kr_index_ntotal) BIND(kr_index_get_rows)
    BIND(kr_index_search) BIND(kr_index_search_async) BIND(kr_index_search_finish) BIND(kr_index_search_finish_ex) BIND(kr_index_search_pending)
    BIND(kr_index_stats) BIND(kr_index_prepare)
    BIND(kr_encoder_create_ex) BIND(kr_encoder_operand_dtype) BIND(kr_encoder_residual_lo) BIND(kr_encoder_destroy) BIND(kr_encoder_load_weight)
    BIND(kr_encoder_finalize) BIND(kr_encoder_forward) BIND(kr_encoder_forward_tt) BIND(kr_encoder_forward_packed) BIND(kr_encoder_check)
    if (p_kr_abi_version() != KR_ABI_VERSION) { printf("ABI version mismatch\n"); return 1; }
    if (p_kr_device_count() < 1) { printf("no GPU visible\n"); return 3; }

    FILE* f = fopen(argv[2], "rb");
    if (!f) { printf("cannot open %s\n", argv[2]); return 2; }
    char magic[4];
    int32_t dims[4];
    if (rd(f, magic, 4) || memcmp(magic, "KRT1", 4) || rd(f, dims, sizeof dims)) { printf("bad vectors file\n"); return 2; }
    const int n = dims[0], d = dims[1], nq = dims[2], k = dims[3];
    float* x = malloc((size_t)n * d * 4); float* q = malloc((size_t)nq * d * 4);
    float* es = malloc((size_t)nq * k * 4); int64_t* er = malloc((size_t)nq * k * 8);
    if (rd(f, x, (size_t)n * d * 4) || rd(f, q, (size_t)nq * d * 4) || rd(f, es, (size_t)nq * k * 4) || rd(f, er, (size_t)nq * k * 8)) return 2;

    /* ---------------- index: Indexer.__init__ / index_data / search_knn (retriever/index.py:19-24, 32, 47) ---------------- */
    kr_index* ix = NULL;
    CHECK(p_kr_index_create(d, KR_METRIC_INNER_PRODUCT, KR_COARSE_BF16, 0, &ix));
    CHECK(p_kr_index_reserve(ix, n));
    const int n1 = n / 3;
    CHECK(p_kr_index_add(ix, x, n1, NULL));
    CHECK(p_kr_index_add(ix, x + (size_t)n1 * d, n - n1, NULL));
    if (p_kr_index_ntotal(ix) != n) { printf("ntotal %lld != %d\n", (long long)p_kr_index_ntotal(ix), n); return 1; }
    float* s0 = malloc((size_t)nq * k * 4); int64_t* r0 = malloc((size_t)nq * k * 8);
    float* s1 = malloc((size_t)nq * k * 4); int64_t* r1 = malloc((size_t)nq * k * 8);
    CHECK(p_kr_index_search(ix, q, nq, k, s0, r0, 0, NULL));
    if (memcmp(r0, er, (size_t)nq * k * 8) || memcmp(s0, es, (size_t)nq * k * 4)) { printf("blocking search differs from the oracle\n"); return 1; }
    memset(s0, 0, (size_t)nq * k * 4); memset(r0, 0, (size_t)nq * k * 8);
    /* two calls outstanding on one stream, finished together (ABI 5) */
    CHECK(p_kr_index_search_async(ix, q, nq, k, s0, r0, NULL));
    CHECK(p_kr_index_search_async(ix, q, nq, k, s1, r1, NULL));
    if (p_kr_index_search_pending(ix) != 2) { printf("pending %d != 2\n", p_kr_index_search_pending(ix)); return 1; }
    int64_t flagged[4] = {-1, -1, -1, -1}; int ncalls = -1;
    CHECK(p_kr_index_search_finish_ex(ix, flagged, 4, &ncalls));
    if (ncalls != 2 || flagged[0] < 0 || flagged[1] < 0 || p_kr_index_search_pending(ix) != 0) { printf("finish_ex: ncalls %d\n", ncalls); return 1; }
    if (memcmp(r0, er, (size_t)nq * k * 8) || memcmp(s0, es, (size_t)nq * k * 4) || memcmp(r1, er, (size_t)nq * k * 8) || memcmp(s1, es, (size_t)nq * k * 4)) {
        printf("async search differs from the oracle\n"); return 1;
    }
    CHECK(p_kr_index_search_finish(ix));                  /* nothing outstanding: a no-op */
    kr_search_stats st;
    CHECK(p_kr_index_stats(ix, &st, 0));
    if (st.queries != 3 * (int64_t)nq || st.certified + st.fallback != st.queries) { printf("stats: %lld queries\n", (long long)st.queries); return 1; }
    float* back = malloc((size_t)5 * d * 4);
    CHECK(p_kr_index_get_rows(ix, n - 5, 5, back, NULL));
    if (memcmp(back, x + (size_t)(n - 5) * d, (size_t)5 * d * 4)) { printf("get_rows differs\n"); return 1; }
    p_kr_index_destroy(ix);

    /* ---------------- encoder: E5Encoder.forward (retriever/encoders.py:67-77) ---------------- */
    int32_t ci[7]; float eps;
    if (rd(f, ci, sizeof ci) || rd(f, &eps, 4)) return 2;
    kr_bert_cfg cfg = {ci[0], ci[1], ci[2], ci[3], ci[4], ci[5], ci[6], eps};
    int32_t nt;
    if (rd(f, &nt, 4)) return 2;
    char** names = malloc(sizeof(char*) * nt); float** data = malloc(sizeof(float*) * nt); int64_t* numel = malloc(8 * nt);
    for (int i = 0; i < nt; ++i) {
        int32_t nl;
        if (rd(f, &nl, 4)) return 2;
        names[i] = calloc(nl + 1, 1);
        if (rd(f, names[i], nl) || rd(f, &numel[i], 8)) return 2;
        data[i] = malloc((size_t)numel[i] * 4);
        if (rd(f, data[i], (size_t)numel[i] * 4)) return 2;
    }
    int32_t bs[3];
    if (rd(f, bs, sizeof bs)) return 2;
    const int B = bs[0], S = bs[1], pool = bs[2], H = cfg.hidden;
    int64_t* ids = malloc((size_t)B * S * 8); int64_t* mask = malloc((size_t)B * S * 8);
    float* eemb = malloc((size_t)B * H * 4); float tol;
    if (rd(f, ids, (size_t)B * S * 8) || rd(f, mask, (size_t)B * S * 8) || rd(f, eemb, (size_t)B * H * 4) || rd(f, &tol, 4)) return 2;
    fclose(f);
    float* emb = malloc((size_t)B * H * 4);
    const int modes[2][2] = {{KR_ENC_F16, 1}, {KR_ENC_BF16, 0}};
    for (int m = 0; m < 2; ++m) {
        kr_encoder* enc = NULL;
        CHECK(p_kr_encoder_create_ex(&cfg, 0, modes[m][0], modes[m][1], &enc));
        if (p_kr_encoder_operand_dtype(enc) != modes[m][0] || p_kr_encoder_residual_lo(enc) != modes[m][1]) { printf("create_ex ignored its arguments\n"); return 1; }
        if (p_kr_encoder_forward(enc, ids, mask, B, S, pool, emb, NULL) != KR_ESTATE) { printf("forward before finalize must fail\n"); return 1; }
        for (int i = 0; i < nt; ++i) CHECK(p_kr_encoder_load_weight(enc, names[i], data[i], numel[i]));
        CHECK(p_kr_encoder_finalize(enc));
        CHECK(p_kr_encoder_forward(enc, ids, mask, B, S, pool, emb, NULL));
        CHECK(p_kr_encoder_check(enc));
        double worst = 0.0;
        for (size_t i = 0; i < (size_t)B * H; ++i) {
            const double e = fabs((double)emb[i] - (double)eemb[i]);
            if (!(e <= worst)) worst = e;                 /* NaN-propagating max */
        }
        const double bar = modes[m][0] == KR_ENC_F16 ? tol : 4.0 * tol;
        if (!(worst <= bar)) { printf("encoder mode %d: max |err| %.3e > %.3e\n", m, worst, bar); return 1; }
        printf("encoder mode (dtype %d, lo %d): max |err| %.2e\n", modes[m][0], modes[m][1], worst);
        /* token types: NULL and all-zero types are kr_encoder_forward bit for bit */
        {
            float* emb2 = malloc((size_t)B * H * 4); int64_t* tt0 = calloc((size_t)B * S, 8);
            CHECK(p_kr_encoder_forward_tt(enc, ids, mask, NULL, B, S, pool, emb2, NULL));
            if (memcmp(emb, emb2, (size_t)B * H * 4)) { printf("forward_tt(NULL) differs\n"); return 1; }
            CHECK(p_kr_encoder_forward_tt(enc, ids, mask, tt0, B, S, pool, emb2, NULL));
            if (memcmp(emb, emb2, (size_t)B * H * 4)) { printf("forward_tt(zeros) differs\n"); return 1; }
            free(emb2); free(tt0);
        }
        /* ragged input (kr_encoder_forward_packed): when the fixture's masks are right-padded, the attended ids back to back + one length per sequence give
         * the same rows bit for bit; lengths that do not add up are KR_EINVAL */
        {
            int32_t* rag = malloc((size_t)B * S * 4); int32_t* lens = malloc((size_t)B * 4); int64_t T = 0; int right = 1;
            for (int b = 0; b < B; ++b) {
                int n = 0;
                for (int p = 0; p < S; ++p) {
                    if (mask[(size_t)b * S + p]) { if (p != n) right = 0; rag[T + n] = (int32_t)ids[(size_t)b * S + p]; ++n; }
                }
                lens[b] = n; T += n;
            }
            if (right) {
                float* emb2 = malloc((size_t)B * H * 4);
                CHECK(p_kr_encoder_forward_packed(enc, rag, lens, B, S, T, pool, emb2, NULL));
                if (memcmp(emb, emb2, (size_t)B * H * 4)) { printf("forward_packed differs from forward\n"); return 1; }
                if (p_kr_encoder_forward_packed(enc, rag, lens, B, S, T - 1, pool, emb2, NULL) != KR_EINVAL || !strstr(p_kr_last_error(), "seq_lens")) {
                    printf("inconsistent lengths not reported: %s\n", p_kr_last_error()); return 1;
                }
                free(emb2);
                printf("forward_packed: bit-identical to forward (%lld tokens of %d x %d)\n", (long long)T, B, S);
            } else printf("forward_packed: fixture is not right-padded, skipped\n");
            free(rag); free(lens);
        }
        /* a token id outside the vocabulary: reported by the host-output call itself */
        const int64_t keep = ids[1];
        ids[1] = cfg.vocab + 7;
        if (p_kr_encoder_forward(enc, ids, mask, B, S, pool, emb, NULL) != KR_EINVAL || !strstr(p_kr_last_error(), "token id")) {
            printf("out-of-vocabulary id not reported: %s\n", p_kr_last_error()); return 1;
        }
        ids[1] = keep;
        CHECK(p_kr_encoder_forward(enc, ids, mask, B, S, pool, emb, NULL));   /* the handle is usable again */
        p_kr_encoder_destroy(enc);
    }
    printf("capi_gpu ok\n");
    dlclose(h);
    return 0;
}
