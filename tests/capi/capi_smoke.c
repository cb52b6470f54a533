/* Plain-C (C99) consumer of include/kirag_amd.h: proves the boundary is a C ABI (no C++ / torch types), binds the library the way a
 * non-Python host would (dlopen + dlsym against the header's prototypes) and exercises the entry points that need no GPU.
 * Built and run by tests/test_capi_c.py:  gcc -std=c99 -Wall -Wextra -pedantic -Iinclude tests/capi/capi_smoke.c -ldl */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "kirag_amd.h"

#define BIND(name)                                               \
    __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); \
    if (!p_##name) { printf("missing symbol %s\n", #name); return 2; }

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: capi_smoke /path/to/libkirag_amd.so\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen failed: %s\n", dlerror()); return 2; }
    /* EVERY entry point include/kirag_amd.h declares (tests/test_capi_c.py checks this list against the header) */
    BIND(kr_abi_version) BIND(kr_last_error) BIND(kr_device_count) BIND(kr_set_option) BIND(kr_release_scratch) BIND(kr_index_create)
    BIND(kr_index_destroy) BIND(kr_index_reserve) BIND(kr_index_add) BIND(kr_index_ntotal) BIND(kr_index_dim) BIND(kr_index_get_rows)
    BIND(kr_index_coarse_dim) BIND(kr_index_coarse_dtype) BIND(kr_index_get_coarse) BIND(kr_index_get_bounds) BIND(kr_index_add_raw)
    BIND(kr_index_search) BIND(kr_index_search_async) BIND(kr_index_search_finish) BIND(kr_index_search_finish_ex) BIND(kr_index_search_finish_one) BIND(kr_index_search_pending)
    BIND(kr_index_stats) BIND(kr_index_prepare) BIND(kr_score_topk) BIND(kr_topk_merge) BIND(kr_format_ids) BIND(kr_index_search_coarse_async) BIND(kr_index_search_global_theta) BIND(kr_index_search_rerank_async) BIND(kr_topk_merge_device) BIND(kr_comm_unique_id) BIND(kr_comm_create)
    BIND(kr_comm_destroy) BIND(kr_comm_rank) BIND(kr_comm_world) BIND(kr_shard_allgather_topk) BIND(kr_encoder_create) BIND(kr_encoder_create_ex)
    BIND(kr_encoder_operand_dtype) BIND(kr_encoder_residual_lo) BIND(kr_encoder_destroy) BIND(kr_encoder_load_weight) BIND(kr_encoder_finalize)
    BIND(kr_encoder_forward) BIND(kr_encoder_forward_tt) BIND(kr_encoder_forward_packed) BIND(kr_encoder_check) BIND(kr_encoder_last_hidden)
    if (p_kr_abi_version() != KR_ABI_VERSION) { printf("ABI version mismatch\n"); return 1; }

    /* argument validation happens before any device work */
    kr_index* ix = NULL;
    if (p_kr_index_create(63, KR_METRIC_INNER_PRODUCT, KR_COARSE_BF16, 0, &ix) != KR_EINVAL || ix != NULL) { printf("bad d accepted\n"); return 1; }
    if (!strstr(p_kr_last_error(), "vector size")) { printf("unexpected message: %s\n", p_kr_last_error()); return 1; }
    kr_bert_cfg cfg = {100, 2, 2, 512, 100, 64, 2, 1e-12f};   /* hidden not a multiple of 128 */
    kr_encoder* enc = NULL;
    if (p_kr_encoder_create(&cfg, 0, &enc) != KR_EINVAL) { printf("bad config accepted\n"); return 1; }
    if (p_kr_index_ntotal(NULL) != 0 || p_kr_index_dim(NULL) != 0) return 1;
    p_kr_index_destroy(NULL);
    p_kr_encoder_destroy(NULL);
    if (p_kr_set_option("no_such_option", 1) != KR_EINVAL || p_kr_set_option("force_exact_scores", 0) != KR_OK) { printf("kr_set_option\n"); return 1; }
    p_kr_release_scratch();   /* nothing allocated yet: must be a no-op */

    /* host-side k-way merge of two shards (score desc, id asc), with a tie across shards and a short shard (-1 padding) */
    const float s[2][1][3] = {{{0.9f, 0.5f, 0.1f}}, {{0.9f, 0.7f, -INFINITY}}};
    const int64_t id[2][1][3] = {{{4, 2, 9}}, {{1, 8, -1}}};
    float os[3]; int64_t oi[3];
    if (p_kr_topk_merge(&s[0][0][0], &id[0][0][0], 2, 1, 3, os, oi) != KR_OK) { printf("merge failed: %s\n", p_kr_last_error()); return 1; }
    if (!(oi[0] == 1 && oi[1] == 4 && oi[2] == 8 && os[0] == 0.9f && os[1] == 0.9f && os[2] == 0.7f)) {
        printf("merge result wrong: %lld %lld %lld\n", (long long)oi[0], (long long)oi[1], (long long)oi[2]);
        return 1;
    }
    /* ids -> decimal ASCII (the bulk form of Indexer.search_knn's str(id) per hit) */
    {
        const int64_t v[4] = {0, -7, 1234567890123LL, INT64_MIN};
        char buf[96]; int64_t n = 0;
        if (p_kr_format_ids(v, 4, ' ', buf, sizeof buf, &n) != KR_OK || n != 39 || memcmp(buf, "0 -7 1234567890123 -9223372036854775808", 39) != 0) {
            printf("kr_format_ids wrong: %.*s\n", (int)n, buf); return 1; }
        if (p_kr_format_ids(v, 4, ' ', buf, 38, &n) != KR_EINVAL) { printf("kr_format_ids accepted a short buffer\n"); return 1; }
    }
    printf("capi_smoke ok (devices visible: %d)\n", p_kr_device_count());
    dlclose(h);
    return 0;
}
