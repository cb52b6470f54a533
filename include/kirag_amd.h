/*
 * kirag_amd — C ABI of the MI355X-native dense-retrieval core (libkirag_amd.so).
 *
 * The reference (jyfang6/kirag) is pure Python and has no FFI of its own: its boundary for this path is the
 * duck-typed Python surface of retriever/{encoders,index,retrievers,e5}.py, underneath which the arithmetic is
 * done by third-party libraries (HF BertModel.forward, faiss.IndexFlatIP).  Each entry point below names the
 * reference call it replaces (paths relative to the reference root).  The Python shims in kirag_amd/ bind these
 * through ctypes; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative KR_E* code; it never throws across the ABI;
 *     kr_last_error() returns a thread-local message for the last failure on the calling thread.
 *   - data pointers may be HOST or DEVICE pointers (hipMemcpyDefault semantics); device pointers must belong to
 *     the handle's device.  The caller owns every I/O buffer; handles own their device allocations.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls on one handle are serialised
 *     by the caller (one thread per rank, as in the reference).
 *   - there is NO CPU fallback: without a gfx950 device every compute call fails with KR_ENODEV.
 */
#ifndef KIRAG_AMD_H
#define KIRAG_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KR_OK 0
#define KR_EINVAL (-22)   /* bad argument (shape, k, dtype, unknown weight name ...) */
#define KR_ENOMEM (-12)   /* hipMalloc failed */
#define KR_ENODEV (-19)   /* no usable HIP device */
#define KR_EHIP (-5)      /* a HIP runtime call failed; see kr_last_error() */
#define KR_ESTATE (-1)    /* handle not ready (e.g. encoder weights missing) */
#define KR_ERANGE (-34)   /* the encoder met non-finite activations (a value outside the f16 operand range, or NaN / Inf weights): results unusable */

#define KR_ABI_VERSION 9
int kr_abi_version(void);
const char* kr_last_error(void);
int kr_device_count(void);
/* process-wide test / diagnostic switches.  "force_exact_scores" (0/1): every canonical score goes through the integer
 * super-accumulator instead of the certified fp64 fast path (same results by definition; exercises the rare path).
 * "byte_prescan" (0/1, default 1): blocks of at most 32 queries on an index of >= 2^19 rows (256 < d <= 1024 with a 16-bit row pitch of 384 / 512 / 768 / 1024) stream an int8 copy of the rows
 * (1 KiB per row at d = 1024, built by the first such search) in the final round of the coarse scan and 16-bit-score only the rows it marks;
 * results are the same exact top-k either way.  KIRAG_AMD_NO_BYTE_SCAN in the environment at kr_index_create: never for that index.
 * "debug_byte_min_rows" (rows; < 0 = default 2^19): test hook, the index size from which small blocks take that path.
 * "debug_eps8_permille" (default 1000; <= 0 restores it): MUTATION hook of the tests — the pre-scan's error bound eps8 is multiplied by value / 1000.  With a value
 * below 1000 the pre-scan is no longer exact (tests/test_gpu_search.py builds a row that a 750-permille bound misses and the real bound finds); never set it in production.
 * Regimes the pre-scan does not serve: blocks of 33-128 queries, and 16-32-query blocks over anisotropic (e5-like) rows that mark more than n/8 rows four times
 * in a row (the index then pauses the pre-scan for 1024 calls): those run the 2-byte stream at ~0.66-0.69 of HBM.  No reference caller sits there
 * (knowledge_graph/models.py:1645 searches 1-2 queries per hop, retrieve.py whole query sets).
 * "debug_va_retired_tib" (TiB) / "debug_vmm_min_reserve_mib" (MiB; 0 = default): test hooks of the large-index address-space
 * budget and of the smallest address range reserved per large index.  Unknown names: KR_EINVAL. */
int kr_set_option(const char* name, int value);
/* frees the per-device scratch buffers kr_score_topk keeps between calls */
void kr_release_scratch(void);

/* ------------------------------------------------------------------------------------------------------------
 * Flat inner-product index — replaces faiss.IndexFlatIP behind retriever/index.py (Indexer, :17-83).
 * Stores, per row, the fp32 master (exact re-rank) and a 16-bit copy (MFMA coarse scan); rows live in HBM.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct kr_index kr_index;

#define KR_METRIC_INNER_PRODUCT 0  /* retriever/index.py:13  FAISSINDEX_DICT["inner_product"] */
#define KR_COARSE_BF16 0
#define KR_COARSE_F16 1

/* Indexer.__init__ (index.py:19-24): faiss.IndexFlatIP(vector_sz).  d % 4 == 0, 4 <= d <= 4096. */
int kr_index_create(int d, int metric, int coarse_dtype, int device, kr_index** out);
void kr_index_destroy(kr_index* ix);
/* capacity hint (rows); avoids regrowth copies when the final size is known (faiss_index_corpus.py:42-45 loop) */
int kr_index_reserve(kr_index* ix, int64_t n_rows);
/* Indexer.index_data -> index.add(embeddings.astype('float32')) (index.py:26-34): append n fp32 rows [n,d]. */
int kr_index_add(kr_index* ix, const float* x, int64_t n, void* stream);
/* index.ntotal (index.py:75,79) */
int64_t kr_index_ntotal(const kr_index* ix);
int kr_index_dim(const kr_index* ix);
/* read back fp32 rows [start, start+n) (serialisation: index.py:55-64 writes the flat storage) */
int kr_index_get_rows(kr_index* ix, int64_t start, int64_t n, float* out, void* stream);

/* Native shard files (SURVEY 8f-1 "native sharded bf16 format"; kirag_amd/retriever/index.py ShardedIndexer.serialize /
 * deserialize_from, next to the reference's index.py:55-79 formats): the stored state of rows, exactly — fp32 master, 16-bit
 * scan copy [n, coarse_dim] and the two quantisation bounds — so that a saved shard reloads without re-quantising. */
int kr_index_coarse_dim(const kr_index* ix);      /* row length of the 16-bit copy (d rounded up to 64) */
int kr_index_coarse_dtype(const kr_index* ix);    /* KR_COARSE_BF16 / KR_COARSE_F16 */
int kr_index_get_coarse(kr_index* ix, int64_t start, int64_t n, uint16_t* out, void* stream);
int kr_index_get_bounds(kr_index* ix, float* out2 /* max |x - c(x)|, max |c(x)| over the stored rows */);
int kr_index_add_raw(kr_index* ix, const float* xf, const uint16_t* xc, int64_t n, const float* bounds2, void* stream);

/* Indexer.search_knn -> index.search(q, top_docs) (index.py:47): exact inner-product top-k.
 *   q       [nq,d] fp32;  scores [nq,k] fp32 (descending);  rows [nq,k] int64 = internal row numbers
 *   (the caller maps them through index_id_to_db_id exactly as index.py:49 does).
 * Result definition (identical to oracle/search_c.c): score = the EXACT inner product of the fp32 inputs rounded once to
 * fp32 (round-to-nearest-even; independent of any summation order); ranking by (score desc, row asc); rows whose score is
 * NaN are never returned.  0 < k <= min(ntotal, 1024), else KR_EINVAL.
 * `mode`: 0 = auto: pass 1 = 16-bit MFMA scan + certified exact re-rank for all queries; pass 2 = fp64 MFMA scan of the fp32 rows
 *             + certified re-rank for the queries pass 1 could not certify (they share one pass over the corpus per group of
 *             32); pass 3 = exact scan, query by query, for what is left (mass ties);
 *         1 = exact scan only, 2 = pass 2 (+3) only — slow; used by tests as on-device cross-checks.
 * Small calls (ABI 8, nq <= 32, mode 0): when q, scores and rows are all addressable by the device (device memory of the index's GPU, pinned host
 * memory) the kernels read the queries and write the results in place — no staging copies; any other combination goes through the workspace as before.
 * On an index of >= 2^19 rows the final coarse round of such a call streams an int8 copy of the rows (kr_set_option "byte_prescan"). */
int kr_index_search(kr_index* ix, const float* q, int nq, int k, float* scores, int64_t* rows, int mode, void* stream);
/* The same search (mode 0) in two halves.  kr_index_search_async ENQUEUES pass 1 of every 1024-query block on `stream` and returns without waiting
 * for the device (given device pointers it performs no host synchronisation at all): the results of every query whose exactness certificate holds
 * are written to scores / rows in stream order, and the per-query certificate flags are copied to pinned memory behind them.
 * kr_index_search_finish waits for that point, reads the flags and - only for queries pass 1 could not certify - runs passes 2 / 3 and overwrites
 * their rows; after it returns the results are final and the statistics are updated.  q, scores and rows must stay valid until then.
 * Up to 16 calls may be outstanding per handle as long as they use ONE stream (ABI 5; the row-sharded search of bench.py --gpus N enqueues the W
 * batches of a block back to back and looks at the certificates once): a call on another stream, a 17th call, and any other call that touches the
 * handle's rows (add, reserve, get_rows ...) finish the outstanding ones first.  kr_index_search_finish finishes ALL outstanding calls, oldest first;
 * kr_index_search_finish_ex does the same and reports, per call (oldest first, the first `cap` of them), how many of its queries pass 1 could not
 * certify — i.e. whether rows of that call's result buffers were re-written after whatever the caller enqueued behind the call had consumed them.
 * kr_index_search(mode 0) == kr_index_search_async + kr_index_search_finish. */
int kr_index_search_async(kr_index* ix, const float* q, int nq, int k, float* scores, int64_t* rows, void* stream);
int kr_index_search_finish(kr_index* ix);
int kr_index_search_finish_ex(kr_index* ix, int64_t* flagged, int cap, int* ncalls);
/* ... and only the OLDEST outstanding call (ABI 8): a host that converts block i's results while block i + 1 is already being searched keeps two calls
 * in flight and finishes them one at a time (Indexer.search_knn).  *flagged (may be NULL) as in kr_index_search_finish_ex; no call outstanding: 0, nothing done. */
int kr_index_search_finish_one(kr_index* ix, int64_t* flagged);
int kr_index_search_pending(const kr_index* ix);   /* number of outstanding asynchronous calls */

/* Row-sharded search with the exchange BEFORE the re-rank (replaces the gather of utils/utils.py:145-155 together with kr_topk_merge_device /
 * kr_shard_allgather_topk).  Every shard certifies and re-ranks its OWN top-k in kr_index_search_async: ~0.3 ms of scattered fp32 row gathers per 1000-query
 * batch on every rank whatever the number of shards, although only ~k / W of a shard's rows reach the global top-k.  Split form, one block of nq <= 1024 queries,
 * all three calls enqueue-only on ONE stream with nothing else on the handle in between:
 *   kr_index_search_coarse_async   pass 1's coarse scan; topk [nq, k + 1] (DEVICE memory) receives the k best coarse scores of this shard per query
 *                                  (unsorted, -inf where the shard has fewer candidates) followed by the query's error bound on this shard;
 *   [the host gathers the shards' topk blocks: gathered [nshards][nq][k + 1], rank order irrelevant]
 *   kr_index_search_global_theta   theta[q] = (k-th best coarse score of ALL shards) - (largest error bound of any shard + this shard's): a row of this shard
 *                                  with a coarse score below it cannot be in the global exact top-k;
 *   kr_index_search_rerank_async   certificate + exact re-rank of the candidates above max(local bound, theta); scores / rows [nq, k] as in
 *                                  kr_index_search_async, but a query may get FEWER than k rows: the tail is (-inf, -1) (kr_topk_merge* treat id < 0 as padding).
 *                                  theta == NULL: exactly kr_index_search_async's result.
 * The call is outstanding from the first half on (kr_index_search_finish* as usual; queries pass 1 could not certify are re-answered with the shard's own
 * exact top-k, which merges just as well); a first half whose second half never comes makes the next finish return KR_ESTATE. */
int kr_index_search_coarse_async(kr_index* ix, const float* q, int nq, int k, float* topk, void* stream);
int kr_index_search_global_theta(kr_index* ix, const float* gathered, int nshards, float* theta, void* stream);
int kr_index_search_rerank_async(kr_index* ix, const float* theta, float* scores, int64_t* rows, void* stream);

typedef struct {
    int64_t queries;          /* queries answered since creation / last reset */
    int64_t certified;        /* answered by the fast path with the exactness certificate holding */
    int64_t fallback;         /* not certified by pass 1 (= fine + exact) */
    int64_t overflow;         /* candidate-buffer overflows in pass 1 (subset of fallback) */
    int64_t reranked_rows;    /* fp32 rows gathered by the re-rank kernel */
    int64_t coarse_rounds;    /* coarse GEMM launches */
    double last_coarse_ms;    /* device time of the coarse launches of the last search call (HIP events) */
    double last_total_ms;     /* device time of the whole last search call */
    int64_t fine;             /* answered by pass 2 (fp64 MFMA scan, certified) */
    int64_t exact;            /* answered by pass 3 (exact scan) */
    int64_t fine_rounds;      /* pass-2 scan launches */
    double last_fine_ms;      /* device time of pass 2 in the last search call */
    int64_t marked_passes;    /* pass-2 groups that were pre-scanned (16-bit stream marking the rows the fp64 scan has to visit) */
    int64_t marked_rows;      /* rows marked by those pre-scans, summed over the groups */
    int64_t va_retired_bytes; /* process-wide: virtual addresses retired by released / moved indexes (never reused) */
    int64_t grow_mode;        /* this index: -1 undecided (< 256 MiB), 0 hipMalloc + copy-on-grow, 1 chunks mapped into a reserved address range */
    int64_t byte_scans;       /* query blocks whose final coarse round went through the int8 copy (kr_set_option "byte_prescan") */
    int64_t byte_marked_rows; /* rows those pre-scans marked (then scored from the 16-bit copy), summed */
    int64_t byte_rows;        /* this index: rows its int8 copy currently covers (0: none - never built, released, or not affordable; kr_index_prepare) */
} kr_search_stats;
int kr_index_stats(kr_index* ix, kr_search_stats* out, int reset);
/* Do now what the FIRST search of blocks of `nq` queries x top-`k` would otherwise do on the spot: allocate that shape's search workspaces, set the kernels'
 * function attributes and - when such blocks take the byte pre-scan on this index (kr_set_option "byte_prescan") - build the int8 copy of the rows (8.6 ms at
 * 5M x 1024 rows + 1 KiB per row of HBM; extended by the rows added since an earlier call) with its row bitmap.  The copy is only built while it leaves
 * max(2 GiB, 1/16 of the device) free and is released again when kr_index_add needs the memory (derived data: results never depend on it).  Called by
 * Indexer.index_data / deserialize_from (retriever/index.py:26-34,66-79) so that the first KiRAG hop after a load (knowledge_graph/models.py:1645) costs what
 * every later one costs.  Enqueues on `stream`; waits for searches in flight on the handle.  An empty index: no-op. */
int kr_index_prepare(kr_index* ix, int nq, int k, void* stream);

/* Exact top-k of q . x^T for a small transient candidate set — replaces the torch.matmul + torch.topk of the KiRAG loop's aligner step
 * (knowledge_graph/models.py:1532-1538: [1-2 queries] x [T triples], top-20) and the matmul + argsort of the exemplar / dev-MRR ranking
 * (models.py:1315-1316, kg_generator.py:119-120, trainer/aligner_trainer.py:112-113).
 *   q [nq,d], x [n,d] fp32 (host or device pointers), scores [nq,k] fp32 descending, rows [nq,k] int64 row numbers of x.
 * Same result definition as kr_index_search (canonical score, ties by row asc); 0 < k <= min(n, 1024), nq <= 65535. */
int kr_score_topk(const float* q, int nq, const float* x, int64_t n, int d, int k, float* scores, int64_t* rows, int device, void* stream);

/* Host-side final merge of per-shard results (north_star: "host-side final merge" after the RCCL all-gather).
 *   scores [nshards,nq,k], ids [nshards,nq,k] (GLOBAL ids) -> out_scores/out_ids [nq,k] by (score desc, id asc).
 *   Host pointers only. */
int kr_topk_merge(const float* scores, const int64_t* ids, int nshards, int nq, int k, float* out_scores, int64_t* out_ids);

/* Decimal ASCII of n int64 ids, joined by `sep`, no trailing separator: the bulk form of the `str(id)` per hit that Indexer.search_knn returns
 * (index.py:49 builds 100 k Python strings per 1024-query x top-100 block one at a time — as long as the GPU's whole search of the block; the
 * host mirror splits this buffer instead).  Host pointers.  cap >= 21 * n always suffices; KR_EINVAL when `out` is too small. */
int kr_format_ids(const int64_t* ids, int64_t n, char sep, char* out, int64_t cap, int64_t* written);

/* The same merge on the device, for lists that are already in HBM (the output of the all-gather): asynchronous on `stream`, no host round trip.
 *   scores + s * score_shard_stride -> list block [nq,k] of shard s (strides in elements), likewise ids; out_* [nq,k] device (or pinned host) pointers.
 *   nshards * k <= 8192.  Result identical to kr_topk_merge (lists sorted by (score desc, id asc), id < 0 = padding at the tail, ids unique). */
int kr_topk_merge_device(const float* scores, int64_t score_shard_stride, const int64_t* ids, int64_t id_shard_stride, int nshards, int nq, int k,
                         float* out_scores, int64_t* out_ids, int device, void* stream);

/* The exchange step of the row-sharded search for hosts without their own collective library (the reference gathers to rank 0 with
 * torch.distributed, utils/utils.py:145-155): one process per GPU, every rank holds rows [row_offset, row_offset + ntotal) of the corpus.
 *   kr_comm_unique_id  rank 0 fills a KR_COMM_ID_BYTES buffer and hands it to the other ranks by any means (file, socket, MPI, torch.distributed ...).
 *   kr_comm_create     collective: blocks until all `world` ranks (one per GPU of the node) have called it with the same id.  RCCL is loaded at run time
 *                      (librccl.so.1, override with KIRAG_AMD_RCCL_LIB); KR_ESTATE if it cannot be loaded.
 *   kr_shard_allgather_topk   collective, enqueue-only on `stream`: all-gather (RCCL over xGMI) of every rank's [nq,k] lists (scores fp32 descending, rows
 *                      int64 GLOBAL row numbers, (-inf, -1) padding at the tail) + the device merge; every rank gets the global top-k in out_* (device
 *                      pointers, [nq,k]).  world * k <= 8192.  Same result as kr_topk_merge of the W lists. */
#define KR_COMM_ID_BYTES 128
typedef struct kr_comm kr_comm;
int kr_comm_unique_id(void* id128);
int kr_comm_create(const void* id128, int rank, int world, int device, kr_comm** out);
int kr_comm_destroy(kr_comm* c);
int kr_comm_rank(const kr_comm* c);
int kr_comm_world(const kr_comm* c);
int kr_shard_allgather_topk(kr_comm* c, const float* scores_local, const int64_t* rows_local, int nq, int k, float* out_scores, int64_t* out_rows,
                            void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * BERT-family sentence encoder — replaces HF BertModel.forward + pooling + F.normalize behind
 * retriever/encoders.py (E5Encoder.forward :67-77, BGEEncoder.forward :106-118) and retriever/e5.py:51-61.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct kr_encoder kr_encoder;

typedef struct {
    int hidden;        /* H   (1024 for e5-large-v2 / bge-large-en) */
    int layers;        /* L   (24) */
    int heads;         /* 16; hidden / heads must be 64 */
    int intermediate;  /* FF  (4096) */
    int vocab;         /* 30522 */
    int max_pos;       /* 512 */
    int type_vocab;    /* 2 */
    float ln_eps;      /* 1e-12 */
} kr_bert_cfg;

#define KR_POOL_MEAN 0 /* E5: average_pool (encoders.py:56-58) then F.normalize */
#define KR_POOL_CLS 1  /* BGE: last_hidden[:,0] (encoders.py:116) then F.normalize */

/* 16-bit type of the MFMA operands and of every stored activation (weights, residual stream, q / k / v, P, ctx, h); accumulation is fp32 and the
 * LayerNorms run in fp32 either way.  f16 has 11 significand bits (|x| <= 65504), bf16 8: with outlier hidden channels (what real BERT-family
 * checkpoints have) only f16 + the residual low half stays inside the 1e-3 score tolerance with margin (tests/golden G10). */
#define KR_ENC_BF16 0
#define KR_ENC_F16 1
#define KR_ENC_DEFAULT_DTYPE KR_ENC_F16
#define KR_ENC_DEFAULT_RESIDUAL_LO 1
int kr_encoder_create(const kr_bert_cfg* cfg, int device, kr_encoder** out);   /* = kr_encoder_create_ex(cfg, device, -1, -1, out) */
/* operand_dtype: KR_ENC_BF16 / KR_ENC_F16 / -1 = default (environment KIRAG_AMD_ENCODER_DTYPE=bf16|f16, else KR_ENC_DEFAULT_DTYPE);
 * residual_lo: 1 = the residual stream between layers keeps a second 16-bit word (the remainder: 16+ significand bits together), 0 = one word,
 * -1 = default (environment KIRAG_AMD_RESIDUAL_LO=0|1, else KR_ENC_DEFAULT_RESIDUAL_LO). */
int kr_encoder_create_ex(const kr_bert_cfg* cfg, int device, int operand_dtype, int residual_lo, kr_encoder** out);
int kr_encoder_operand_dtype(const kr_encoder* enc);   /* what the handle was created with */
int kr_encoder_residual_lo(const kr_encoder* enc);
void kr_encoder_destroy(kr_encoder* enc);
/* one call per HF state_dict tensor of BertModel ("embeddings.word_embeddings.weight",
 * "encoder.layer.3.attention.self.query.bias", ...), fp32, numel checked; "pooler.*" / "*position_ids" ignored. */
int kr_encoder_load_weight(kr_encoder* enc, const char* hf_name, const float* data, int64_t numel);
/* verifies every tensor was supplied and builds the fused/packed device copies */
int kr_encoder_finalize(kr_encoder* enc);
/* forward(input_ids, attention_mask) (encoders.py:67-77 / :106-118); token_type_ids are 0 as in every caller.
 *   input_ids, attention_mask [B,S] int64 (S <= max_pos), out [B,hidden] fp32 L2-normalised.
 *   A sequence whose mask is all zero yields NaN (mean pool) exactly like the reference. */
int kr_encoder_forward(kr_encoder* enc, const int64_t* input_ids, const int64_t* attention_mask, int B, int S,
                       int pool, float* out, void* stream);
/* The same with token_type_ids [B,S] int64 (HF BertModel.forward's third input; NULL = all zero = kr_encoder_forward).  No KiRAG caller passes non-zero
 * types (the collators encode single texts), but the encoders' forward signature has the argument (encoders.py:67,106).  A value outside
 * [0, type_vocab) is reported as KR_EINVAL through the same deferred channel as token ids. */
int kr_encoder_forward_tt(kr_encoder* enc, const int64_t* input_ids, const int64_t* attention_mask, const int64_t* token_type_ids, int B, int S,
                          int pool, float* out, void* stream);
/* The same forward from RAGGED input: token_ids = int32 ids of the attended positions of every sequence back to back (total_tokens of them, host or device),
 * seq_lens [B] int32 = how many belong to each sequence, occupying positions 0 .. len-1 - what the reference's collator yields for a right-padding tokenizer
 * (dataset/collators.py:59-81, padding=True: input_ids[b, :len], attention_mask[b] = 1^len 0^(S-len)) without the padding: 4-16x fewer bytes from the tokenizer
 * processes of compute_corpus_embeddings.py:77-81 to the GPU.  S = the padded width the equivalent [B,S] call would have (>= every length, <= max_pos; it selects
 * the attention kernel exactly as kr_encoder_forward does).  Rows are BIT-IDENTICAL to kr_encoder_forward on the equivalent padded batch (the same kernels run
 * on the same packed token tables).  A length outside [0, S] or lengths that do not add up to total_tokens are reported as KR_EINVAL through the deferred
 * channel below (the sequence is read as empty, nothing is read out of bounds); an empty sequence yields NaN (mean pool) like an all-zero mask. */
int kr_encoder_forward_packed(kr_encoder* enc, const int32_t* token_ids, const int32_t* seq_lens, int B, int S, int64_t total_tokens, int pool, float* out,
                              void* stream);
/* kr_encoder_forward with a DEVICE `out` pointer only enqueues work on `stream` and returns (no host synchronisation); with a host `out`
 * it returns when the result is in the caller's buffer.  The one thing a forward can get wrong at run time - a token id outside [0, vocab)
 * - is recorded by the kernels (the offending token is read as id 0) and reported as KR_EINVAL by the host-output call itself, or, for
 * device-output calls, by the NEXT call on the handle once that forward has finished, or by kr_encoder_check(), which waits for it.
 * The same channel reports KR_ERANGE when a LayerNorm row of the forward was not finite: with f16 operands (the default) an activation beyond
 * +-65504 becomes inf and the embedding NaN; nothing saturates silently. */
int kr_encoder_check(kr_encoder* enc);
/* debugging / parity: last_hidden_state of the previous forward, fp32 [B*S_packed...] see DESIGN.md */
int kr_encoder_last_hidden(kr_encoder* enc, float* out /* [B,S,hidden] */, int B, int S);

#ifdef __cplusplus
}
#endif
#endif /* KIRAG_AMD_H */
