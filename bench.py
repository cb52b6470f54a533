#!/usr/bin/env python3
"""Headline benchmark of the dense-retrieval hot path on MI355X (BASELINE.json metric: queries/sec + passages-encoded/sec,
e5-large-v2 shape, 5M-doc corpus, top-100).

One "step" = one pass of the hot path over one batch of synthetic input:
    encode a 1k-query batch (32 tokens each) with the BERT-large-shape encoder  +  exact top-100 search of those 1k query
    vectors over the GPU-resident 5M x 1024 corpus (bf16 scan copy + fp32 master rows).
Inputs (token ids, corpus embeddings, weights) are resident in HBM before the timed region.
With --gpus N the SAME job is split over the ranks (strong scaling, SURVEY.md 8e / BASELINE.md 4): the corpus is row-sharded
(5M / N rows per GPU), every rank encodes its 1/N slice of the query batch, the query embeddings are all-gathered (RCCL, nq*4 KiB),
every rank searches all queries on its shard, and the per-shard top-100 are all-gathered and merged on the host.

Prints ONE JSON line on rank 0 (see the round brief for the contract) including
    "roofline"      for the dominant kernel (the MFMA coarse scan k_coarse), timed live with HIP events
    "cpu_baseline"  the reference's CPU path timed on the host cores on a bounded sample of the same step: HF BertModel fp32 query
                    encoding (oracle/encoder_torch.py) + the oracle's fp32 sgemm + top-k stand-in for faiss.IndexFlatIP
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_MFMA_DENSE_16BIT = 2.5e15   # MI355X_MICROARCH.md: bf16/f16 dense MFMA peak
HBM_PEAK = 8.0e12


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--total-rows", type=int, default=5_000_000, help="corpus rows of the whole job (sharded over the ranks)")
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--dim", type=int, default=1024)
    ap.add_argument("--query-tokens", type=int, default=32)
    ap.add_argument("--passages", type=int, default=1024, help="passages per encode-throughput measurement batch")
    ap.add_argument("--passage-tokens", type=int, default=128)
    ap.add_argument("--coarse-dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--no-encoder", action="store_true", help="search-only step (query vectors pre-computed)")
    ap.add_argument("--encoder-dtype", default=None, choices=["bf16", "f16"], help="16-bit operand type of the encoder (default: the library's, f16)")
    ap.add_argument("--residual-lo", default=None, type=int, choices=[0, 1], help="encoder residual stream with / without its low half (default: the library's, 1)")
    ap.add_argument("--collective", default="torch", choices=["torch", "kr_comm"],
                    help="N > 1: exchange of the per-shard result lists through torch.distributed (default) or the library's own RCCL step (kr_shard_allgather_topk)")
    ap.add_argument("--encode-split", default="batch", choices=["batch", "queries"],
                    help="N > 1: 'batch' = up to W consecutive steps form a block: every rank encodes its 1/W slice of EVERY batch of the block in ONE "
                         "forward (a full-size batch per rank and block), one all-gather of the block's query vectors, then the block's searches enqueue-only "
                         "with one host synchronisation per block; 'queries' = every rank encodes 1/W of every batch, step by step (round-2 schedule)")
    ap.add_argument("--search-stream", action="store_true", help="one GPU experiment: enqueue the search of step i on a second stream (overlaps the encode of step i + 1)")
    ap.add_argument("--sync-search", action="store_true", help="one GPU: use the blocking kr_index_search per step instead of search_async + finish")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the small-batch latency block (forwards of the reference's real batch shapes + one KiRAG hop)")
    ap.add_argument("--no-entry-point", action="store_true", help="skip the cal_doc_embeddings-from-text measurement (encode.entry_point)")
    ap.add_argument("--entry-passages", type=int, default=65536, help="synthetic text passages of the encode.entry_point block")
    ap.add_argument("--no-surface", action="store_true", help="skip the reference-surface search measurement (Indexer.search_knn on --surface-queries queries)")
    ap.add_argument("--surface", action="store_true", help="(default on at N = 1) report queries/s of Indexer.search_knn next to the C-ABI search on the same queries")
    ap.add_argument("--surface-queries", type=int, default=4096)
    ap.add_argument("--cpu-sample-rows", type=int, default=1_000_000, help="corpus rows of the CPU search baseline (BASELINE.md: the 1M point)")
    ap.add_argument("--cpu-full-corpus", action="store_true", help="CPU search baseline over ALL corpus rows (the 5M point: ~20 GB of host memory, minutes)")
    ap.add_argument("--cpu-sample-queries", type=int, default=32)
    ap.add_argument("--corpus-dist", default="gaussian", choices=["gaussian", "e5like", "mixed", "neardup"],
                    help="synthetic corpus: iid Gaussian directions (SURVEY 8d) or e5like (shared mean direction + anisotropic remainder: scores in a "
                         "narrow band around 0.75, rank-100 gaps ~5e-5 at 5M rows; kirag_amd/bench_support.py CorpusDist)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="rank launch + process group + barrier / max-over-ranks + the rank-0 JSON line, NO compute and value = null "
                         "(lets the CPU test suite exercise `--gpus N` without a GPU; never a bench result)")
    return ap.parse_args()


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves, one process per GPU, exactly as the
    driver would (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, the launch model of the reference's
    README.md:86 / utils/utils.py:91-98).  This parent has NOT touched the GPU (no torch import, no HIP call): it only starts the launcher as a
    child process, relays its output and exits with its code; rank 0 prints the one JSON line."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(args, q_host, with_encoder, dist_kind, index):
    """faiss.IndexFlatIP stand-in (oracle.search_np.search_sgemm: fp32 BLAS sgemm + argpartition, 1024-query blocks as retriever/index.py:39-47)
    timed on the REAL rows of the benchmark corpus (1M and 5M x 1024 fp32, 1000 queries, top-100): by default the first
    1,000,000 rows of the resident index are copied to the host and searched (one warm-up, median of 3 passes of ~4 s on this box class); the 5M-row
    figure reported next to it is that pass time x (total_rows / 1M) unless --cpu-full-corpus measures it directly (20 GB of host rows, ~2 minutes;
    the committed profiles/r03/cpu_baseline_5M.json is such a run).  Encoder: the HF BertModel fp32 query loop (oracle/encoder_torch.py),
    one warm-up, median of 5."""
    import torch
    from oracle import search_np as S
    rows = args.total_rows if args.cpu_full_corpus else min(args.cpu_sample_rows, args.total_rows)
    xs = np.empty((rows, args.dim), np.float32)
    for s0 in range(0, rows, 250_000):                                 # the same fp32 rows the GPU path searches
        m = min(250_000, rows - s0)
        xs[s0:s0 + m] = index.reconstruct_n(s0, m)
    reps = 3
    S.search_sgemm(q_host, xs, args.topk)                             # warm-up (BLAS threads, page faults)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        S.search_sgemm(q_host, xs, args.topk)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    dt = ts[reps // 2]
    scale = args.total_rows / rows
    search_qps = len(q_host) / (dt * scale)
    out = {"unit": "queries/s", "cores": int(torch.get_num_threads()), "kind": "port", "nproc": os.cpu_count(),
           "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "protocol": f"1 warm-up, median of {reps} (search) / 5 (encode)",
           "search_only_qps": search_qps, "search_runs_s": [round(t, 4) for t in ts],
           "search_points": {f"{rows}_rows_measured_qps": len(q_host) / dt},
           "sample": f"search: {len(q_host)} queries x the first {rows} of the {args.total_rows} corpus rows ({dist_kind}, copied from the resident index) fp32 "
                     f"sgemm+argpartition top-{args.topk}, median {dt:.3f}s per pass"
                     + (f", time scaled x{scale:g} to the full corpus" if scale != 1 else " (the full corpus: measured, not scaled)")}
    del xs
    import glob
    for cand in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "cpu_baseline_5M.json")), reverse=True):   # a committed full-corpus measurement
        with open(cand) as f:
            cj = json.load(f)
        out["search_points"]["5000000_rows_measured_qps_committed_run"] = {"qps": cj.get("search_only_qps"), "source": os.path.relpath(cand, REPO),
                                                                            "cores": cj.get("cores")}
        break
    if with_encoder:
        from oracle import encoder_torch as ET
        enc_qps, enc_dt, enc_ts = ET.time_encode(args.cpu_sample_queries, args.query_tokens, batch=8, repeats=5)
        out["encode_only_qps"] = enc_qps
        out["encode_runs_s"] = [round(t, 3) for t in enc_ts]
        out["value"] = 1.0 / (1.0 / enc_qps + 1.0 / search_qps)       # same step as the GPU side: encode the batch, then search it
        out["sample"] += (f"; encode: {args.cpu_sample_queries} queries x {args.query_tokens} tokens through HF BertModel (BERT-large shape, fp32, "
                          f"batches of 8), median {enc_dt:.2f}s per pass; value = 1/(1/encode + 1/search)")
    else:
        out["value"] = search_qps
    return out


def latency_block(args, encoder, index, dev):
    """Small-batch latency at the batch shapes the reference really calls the path with: one query (e5.py helpers / KiRAG.retrieve,
    knowledge_graph/models.py:1645), one 256-token chain query (models.py:1526-1531), compute_corpus_embeddings' default batch of 8 passages
    (compute_corpus_embeddings.py:43), a triple batch of 125 x 32 — ms per forward, forwards back to back, median of 3 rounds of 30 — and one KiRAG hop on
    this GPU: encode ONE 256-token chain query, exact top-10 over the resident corpus (enqueue both, one host wait)."""
    import torch
    from kirag_amd import bench_support as BS
    out = {"unit": "ms", "forwards": {}}
    for B, S in ((1, 32), (1, 256), (8, 128), (125, 32)):
        ids, mask = BS.synthetic_tokens(dev, B, S, seed=1)
        for _ in range(3):
            encoder.forward(ids, mask, 0)
        rounds = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30):
                encoder.forward(ids, mask, 0)
            torch.cuda.synchronize(); rounds.append((time.perf_counter() - t0) / 30 * 1e3)
        out["forwards"][f"{B}x{S}"] = float(np.median(rounds))
    ids, mask = BS.synthetic_tokens(dev, 1, 256, seed=7)
    k = 10
    sc = torch.empty((1, k), dtype=torch.float32, pin_memory=True); rw = torch.empty((1, k), dtype=torch.int64, pin_memory=True)
    hops = []
    for i in range(23):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        qv = encoder.forward(ids, mask, 0)
        index.search_async(qv, k, sc, rw)
        index.finish()
        if i >= 3:
            hops.append((time.perf_counter() - t0) * 1e3)
    out["kirag_hop_nq1"] = {"ms": float(np.median(hops)), "what": f"encode 1 x 256 tokens + exact top-{k} over {index.ntotal} resident rows, results in pinned host memory"}
    # the search half of the hop alone, with and without the int8 final round of small blocks (kr_set_option "byte_prescan"): same rows, same scores
    from kirag_amd import _lib as L
    qv = encoder.forward(ids, mask, 0)
    alone = {}
    for name, on in (("search_16bit_final_round_ms", 0), ("search_ms", 1)):
        L.check(L.load().kr_set_option(b"byte_prescan", on))
        ts = []
        for i in range(23):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            index.search_async(qv, k, sc, rw); index.finish()
            if i >= 3:
                ts.append((time.perf_counter() - t0) * 1e3)
        alone[name] = float(np.median(ts))
        alone["rows_" + name[:-3]] = rw.clone().numpy().tolist()[0]
    st = index.stats()
    out["kirag_hop_nq1"].update({"search_ms": alone["search_ms"], "search_16bit_final_round_ms": alone["search_16bit_final_round_ms"],
                                 "same_rows_either_way": alone["rows_search"] == alone["rows_search_16bit_final_round"],
                                 "byte_scans": int(st.get("byte_scans", 0)), "byte_marked_rows_per_scan": (st.get("byte_marked_rows", 0) / max(1, st.get("byte_scans", 0)))})
    # the same 1 x 32 forward through the reference's own surface: E5Encoder.forward (retriever/encoders.py:67-77) of an nn.Module with the e5-large shape —
    # what BaseRetriever.query / the e5.py helpers call; one forward at a time (forward + wait), which is how a hop uses it
    from transformers import BertConfig
    from kirag_amd.retriever.encoders import E5Encoder
    c = encoder.cfg if hasattr(encoder, "cfg") else None
    mod = E5Encoder(BertConfig(vocab_size=c.vocab_size, hidden_size=c.hidden_size, num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
                               intermediate_size=c.intermediate_size, max_position_embeddings=c.max_position_embeddings), add_pooling_layer=False).to(dev).eval()
    ids, mask = BS.synthetic_tokens(dev, 1, 32, seed=1)
    for _ in range(3):
        mod(ids, mask)
    torch.cuda.synchronize()
    one = []
    for _ in range(40):
        t0 = time.perf_counter(); mod(ids, mask); torch.cuda.synchronize(); one.append((time.perf_counter() - t0) * 1e3)
    raw = []
    for _ in range(40):
        t0 = time.perf_counter(); encoder.forward(ids, mask, 0); torch.cuda.synchronize(); raw.append((time.perf_counter() - t0) * 1e3)
    out["one_at_a_time_1x32"] = {"module_surface_ms": float(np.median(one)), "c_abi_ms": float(np.median(raw)),
                                 "what": "E5Encoder.forward (nn.Module surface, weight-sync check included) vs HipBertForward.forward, each followed by a device synchronisation"}
    # ONE HOP AT THE REFERENCE'S SURFACE (knowledge_graph/models.py:1645 `self.retriever(queries, topk)` = DenseRetriever.forward -> batch_retrieve,
    # retrievers.py:250-291): a query STRING in, a list of {"id", "score"} out - tokenizer, collator, E5Encoder module forward, Indexer.search_knn, result parsing -
    # over the resident rows, next to the C-ABI pieces (HipBertForward.forward + FlatIPIndex.search) on the same tokens
    import shutil
    import tempfile
    import torch.nn as nn
    from kirag_amd.collators import E5Collator
    from kirag_amd.retriever.index import Indexer
    from kirag_amd.retriever.retrievers import BaseRetriever, DenseRetriever
    td = tempfile.mkdtemp(prefix="kirag_amd_hop_")
    try:
        vocab_file, texts = BS.synthetic_text_corpus(8, td, seed=5)
        col = E5Collator(tokenizer=BS.wordpiece_tokenizer(vocab_file), query_maxlength=256, doc_maxlength=128)

        class Ret(BaseRetriever):
            def __init__(self, encoder):
                nn.Module.__init__(self)
                self.encoder = encoder
                self.norm_query = self.norm_doc = False
                self.temperature, self.local_rank, self.world_size = 1.0, -1, 1
        ixr = Indexer.__new__(Indexer)
        ixr.faiss_padding = False; ixr.index = index
        ixr.index_id_to_db_id = np.arange(index.ntotal, dtype=np.int64) * 3 + 1_000_000
        dr = DenseRetriever(retriever=Ret(mod), collator=col, indexer=ixr, corpus=None, batch_size=4)
        words = texts[0].split()[3:]
        query = "which {} ?\nknowledge triples: {}.".format(" ".join(words[:12]), ". ".join("<" + " ".join(texts[1 + i % 7].split()[3 + i:12 + i]) + ">" for i in range(16)))
        a = col.encode_query([query], max_length=256)
        ntok = int(a["attention_mask"].sum())
        ids, mask = a["input_ids"].to(dev), a["attention_mask"].to(dev)

        def timed(fn, reps=30):
            for _ in range(4):
                fn()
            torch.cuda.synchronize(); ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            return float(np.median(ts))
        res = dr([query], 10)[0]
        qv = mod._hip.forward(ids, mask, 0)
        s_abi, r_abi = index.search(qv, 10)
        same = [d_["id"] for d_ in res] == [str(v) for v in ixr.index_id_to_db_id[r_abi[0]].tolist()]
        if os.environ.get("KIRAG_BENCH_PROFILE_HOP"):          # diagnostic: where the surface hop spends its time in THIS process
            import cProfile
            import pstats
            pr = cProfile.Profile(); pr.enable()
            for _ in range(50):
                dr([query], 10)
            pr.disable()
            pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
        t_surface = timed(lambda: dr([query], 10))
        t_abi = timed(lambda: index.search(mod._hip.forward(ids, mask, 0), 10))
        t_tok = timed(lambda: col.encode_query([query], max_length=256))
        out["kirag_hop_nq1_surface"] = {"ms": t_surface, "c_abi_same_tokens_ms": t_abi, "ratio_to_c_abi": t_surface / t_abi, "tokenizer_collator_ms": t_tok, "query_tokens": ntok,
                                        "same_ids_as_c_abi": bool(same),
                                        "what": "DenseRetriever([query_text], 10) over the resident rows (string in, list of {id, score} out) vs HipBertForward.forward + FlatIPIndex.search on its tokens"}
    finally:
        shutil.rmtree(td, ignore_errors=True)
    del mod
    return out


def entry_point_block(args, encoder, dev):
    """passages-encoded/s at the reference's ENTRY POINT: ``cal_doc_embeddings`` (compute_corpus_embeddings.py:50-134) from passage TEXT with
    DEFAULT flags — tokenizer feed, encoder, append to a resident index shard, shard files written — over synthetic passages in the reference's passage format
    (bench_support.synthetic_text_corpus: 30522-entry WordPiece vocabulary, ~111 tokens per passage), next to the encoder's rate on the SAME token batches
    pre-tokenised and resident in HBM (the figure ``encode.passages_per_s`` is the fixed-length variant of).  ``ratio`` is what the host side keeps of the kernels."""
    import shutil
    import tempfile
    from types import SimpleNamespace
    import torch
    from kirag_amd import bench_support as BS
    from kirag_amd import compute_corpus_embeddings as CC
    from kirag_amd.collators import E5Collator
    from kirag_amd.retriever.index import Indexer
    n, bs = args.entry_passages, 512
    td = tempfile.mkdtemp(prefix="kirag_amd_bench_")
    try:
        vocab_file, texts = BS.synthetic_text_corpus(n, td)
        col = E5Collator(tokenizer=BS.wordpiece_tokenizer(vocab_file), query_maxlength=args.passage_tokens, doc_maxlength=args.passage_tokens)

        class Model:
            def __init__(self): self.encoder = SimpleNamespace(_hip=encoder, config=SimpleNamespace(vocab_size=encoder.cfg.vocab_size))
            def to(self, d): return self
            def eval(self): return self
            def doc(self, a): return encoder.forward(a["input_ids"], a["attention_mask"], 0)
            def doc_packed(self, ids, lens, S, T=None): return encoder.forward_packed(ids, lens, S, 0, T)

        class Corpus:
            def __init__(self, m): self.m = m; self.index_to_passage_id = {i: str(i) for i in range(m)}
            def __len__(self): return self.m
            def __getitem__(self, i): return {"index": i, "passage": texts[i]}
        # the encoder alone on the first 8 of these batches, pre-tokenised and resident
        pre = []
        for s in range(0, min(n, 8 * bs), bs):
            a = col.encode_doc(texts[s:s + bs])
            pre.append((a["input_ids"].to(dev), a["attention_mask"].to(dev)))
        for ids, mask in pre[:2]:
            encoder.forward(ids, mask, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reps = 6
        for _ in range(reps):
            for ids, mask in pre:
                encoder.forward(ids, mask, 0)
        torch.cuda.synchronize()
        pre_rate = reps * sum(len(i) for i, _ in pre) / (time.perf_counter() - t0)
        mean_tokens = float(sum(int(m.sum()) for _, m in pre) / sum(len(i) for i, _ in pre))
        del pre

        def run(m, folder):
            a = CC.setup_parser(["--save_dir", td, "--name", "bench", "--index_folder", folder, "--doc_maxlength", str(args.passage_tokens)])   # every other flag at its default
            ix = Indexer(encoder.hidden)
            t0 = time.perf_counter()
            CC.cal_doc_embeddings(a, Model(), Corpus(m), col, indexer=ix, device=dev)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            files = sorted(os.listdir(os.path.join(td, "bench", folder)))
            assert ix.index.ntotal == m and files == [f"corpus_embeddings_0_{m - 1}.pkl", f"passage_id_list_0_{m - 1}.pkl"], files
            return dt, dict(getattr(CC.cal_doc_embeddings, "last_feed", {}))
        run(min(n, 4 * bs), "warm")                      # first allocations (pinned ring, workspace), tokenizer thread pool
        dt, feed_info = run(n, "timed")
        return {"what": "cal_doc_embeddings from TEXT, default flags (ragged token feed -> kr_encoder_forward_packed -> resident shard + .pkl shard files)",
                "passages": n, "mean_tokens": mean_tokens, "passages_per_s": n / dt, "pretokenised_resident_passages_per_s": pre_rate, "ratio": n / dt / pre_rate,
                "seconds": dt, "feed": feed_info}
    finally:
        shutil.rmtree(td, ignore_errors=True)


def surface_block(args, index, q_vec, dev):
    """queries/s of the call the reference makes — Indexer.search_knn (retriever/index.py:36-53; DenseRetriever.batch_retrieve, retrievers.py:250-275):
    numpy queries in, List[(List[str], np.ndarray)] out — next to the C-ABI search (kr_index_search: all blocks enqueued, one host wait, results in pinned
    memory) on the same queries.  The gap is the reference surface's own host work: 100 k Python strings per 1024-query x top-100 block."""
    import torch
    from kirag_amd.retriever.index import Indexer
    nq, k = args.surface_queries, args.topk
    g = torch.Generator(device=dev); g.manual_seed(11)
    reps = (nq + q_vec.shape[0] - 1) // q_vec.shape[0]
    q = (q_vec.repeat(reps, 1)[:nq] + 0.01 * torch.randn((nq, q_vec.shape[1]), device=dev, generator=g))
    q = torch.nn.functional.normalize(q, dim=1).contiguous()
    q_host = q.cpu().numpy()
    ix = Indexer.__new__(Indexer)
    ix.faiss_padding = False; ix.index = index
    ix.index_id_to_db_id = np.arange(index.ntotal, dtype=np.int64) * 3 + 10_000_000_000
    ps = torch.empty((nq, k), dtype=torch.float32, pin_memory=True); pi = torch.empty((nq, k), dtype=torch.int64, pin_memory=True)
    t_abi, t_surf = [], []
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        index.search_into(q, k, ps, pi)
        t = time.perf_counter() - t0
        res = None                                  # the previous iteration's result (410 k str objects) is freed before the clock starts, not inside search_knn's time
        t0 = time.perf_counter()
        res = ix.search_knn(q_host, k, verbose=False)
        t2 = time.perf_counter() - t0
        if i:
            t_abi.append(t); t_surf.append(t2)
    ok = all(res[r][0] == [str(v) for v in ix.index_id_to_db_id[pi[r].numpy()].tolist()] for r in (0, 1023, 1024, nq - 1))
    a, b = float(np.median(t_abi)), float(np.median(t_surf))
    return {"queries": nq, "topk": k, "search_knn_queries_per_s": nq / b, "c_abi_queries_per_s": nq / a, "ratio": a / b, "search_knn_ms": b * 1e3, "c_abi_ms": a * 1e3,
            "lists_match_c_abi": bool(ok)}


def plumbing_only(args, world, rank):
    """--plumbing-only: everything of the N-rank contract except the GPU work (gloo on CPU)."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "plumbing only (no compute)", "value": None, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                          "dtype": None, "data": "none", "config": {"workload": "none: --plumbing-only"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))               # before any torch / HIP call in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch N ranks with --gpus N (or run `python bench.py --gpus N`, "
                         f"which starts them itself)")
    if args.plumbing_only:
        return plumbing_only(args, world, rank)
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (kirag_amd has no CPU fallback)")
    # one rank per GPU; KIRAG_BENCH_BACKEND=gloo lets a 1-GPU box rehearse the N > 1 code path with several ranks on the same device
    backend = os.environ.get("KIRAG_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from kirag_amd import _lib
    from kirag_amd.retriever.index import FlatIPIndex
    from kirag_amd.parallel import ShardedSearcher

    total, d, nq, k = args.total_rows, args.dim, args.queries, args.topk
    per = (total + world - 1) // world                      # contiguous row shards
    row0 = min(rank * per, total)
    n = min(per, total - row0)                              # rows resident on this rank
    from kirag_amd.bench_support import CorpusDist
    cdist = CorpusDist(args.corpus_dist, d, dev)
    g = torch.Generator(device=dev); g.manual_seed(3 + rank)
    index = FlatIPIndex(d, device=local_rank, coarse_dtype=args.coarse_dtype)
    index.reserve(n)
    chunk = 250_000
    head = None
    for s0 in range(0, n, chunk):            # synthetic unit-norm corpus rows, generated on the device
        m = min(chunk, n - s0)
        x = cdist.rows(m, g)
        if s0 == 0:
            head = x[:nq].clone()
        index.add(x)
        del x
    gq = torch.Generator(device=dev); gq.manual_seed(2)
    # query vectors: near-duplicates of corpus rows of rank 0's generator family (known neighbours on rank 0)
    q_vec = cdist.queries_near(head, gq)
    if world > 1:
        dist.broadcast(q_vec, src=0)

    encoder = None
    q_lo, q_hi = (rank * nq) // world, ((rank + 1) * nq) // world      # this rank's slice of the query batch
    if not args.no_encoder:
        from kirag_amd import bench_support as BS
        encoder = BS.make_hip_encoder(dev, operand_dtype=args.encoder_dtype, residual_lo=None if args.residual_lo is None else bool(args.residual_lo))
        tok_ids, tok_mask = BS.synthetic_tokens(dev, nq, args.query_tokens, seed=2)
        pas_ids, pas_mask = BS.synthetic_tokens(dev, args.passages, args.passage_tokens, seed=1)
    searcher = ShardedSearcher(index, row_offset=row0, world=world, collective=args.collective)
    q_all = torch.empty((nq, d), dtype=torch.float32, device=dev)
    counts = [((r + 1) * nq) // world - (r * nq) // world for r in range(world)]

    # one GPU: the step is enqueue-only (encoder forward -> kr_index_search_async into device buffers -> D2H of the results into pinned memory);
    # the host finishes step i (certificate flags; passes 2 / 3 if any query was flagged) after it has enqueued the encode of step i + 1
    res_s = [torch.empty((nq, k), dtype=torch.float32, device=dev) for _ in range(2)]
    res_i = [torch.empty((nq, k), dtype=torch.int64, device=dev) for _ in range(2)]
    pin_s = [torch.empty((nq, k), dtype=torch.float32, pin_memory=True) for _ in range(2)]
    pin_i = [torch.empty((nq, k), dtype=torch.int64, pin_memory=True) for _ in range(2)]
    inflight, coarse_log = [], []

    def drain():
        if inflight:
            j = inflight.pop()
            index.finish()
            coarse_log.append(index.stats()["last_coarse_ms"])                           # HIP events around the coarse rounds of the finished step
            pin_s[j].copy_(res_s[j], non_blocking=True); pin_i[j].copy_(res_i[j], non_blocking=True)

    side = torch.cuda.Stream(device=dev) if args.search_stream else None

    def step_async(i):
        qv = q_vec if encoder is None else encoder.forward(tok_ids, tok_mask, 0)       # [nq, d] fp32 on the device, enqueued
        drain()
        j = i & 1
        if side is not None:                                                          # experiment: the search of step i on a second stream, under the encode of step i + 1
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                index.search_async(qv, k, res_s[j], res_i[j])
        else:
            index.search_async(qv, k, res_s[j], res_i[j])
        inflight.append(j)
        step_async.keep = qv                                                          # the queries stay alive until finish()

    # N > 1, default schedule ("batch"): up to W consecutive steps form a block.  Rank r ENCODES its 1/W slice of every batch of the block in ONE forward
    # (count x nq / W sequences: with a full block that is a full-size batch, which runs at the large-batch MFMA rate — a 125-query forward does not:
    # 715 vs 1030 TFLOP/s — and a partial last block, steps % W != 0, still gives every rank an equal share instead of idling W - count ranks), ONE
    # all-gather moves the block's query vectors (count x 4 MB over xGMI), then every rank searches the block's batches one after the other on its
    # corpus shard, ENQUEUE ONLY (kr_index_search_async + all-gather of the per-shard top-k + device merge + D2H per batch), and the host synchronises
    # once per block (ShardedSearcher.finish_deferred: certificates of all batches, exchange repeated only for a batch some rank had to re-answer).
    # Same total work and results as one step per batch.  --encode-split queries keeps the round-2 schedule (every rank encodes 1/W of every batch, step by step).
    even = nq % world == 0
    sl = nq // world
    q_gath = torch.empty(world * world * sl * d, dtype=torch.float32, device=dev) if world > 1 and even else None     # gathered slices of a block
    q_blk = torch.empty((world, nq, d), dtype=torch.float32, device=dev) if world > 1 else None                       # [batch-in-block][nq][d]
    rep_ids = rep_mask = None
    if world > 1 and encoder is not None and even:
        rep_ids = tok_ids[q_lo:q_hi].repeat(world, 1).contiguous(); rep_mask = tok_mask[q_lo:q_hi].repeat(world, 1).contiguous()

    def block_of_steps(first, count):
        """steps [first, first + count), count <= W; returns the block's (scores, rows) pinned tensors (valid on return)"""
        if encoder is not None:
            if even:
                # slice `rank` of the block's `count` batches in one forward (the synthetic batches repeat one token batch, as at N = 1)
                mine = encoder.forward(rep_ids[:count * sl], rep_mask[:count * sl], 0)      # [count * sl, d]
                flat = q_gath[:world * count * sl * d].view(world * count * sl, d)          # [rank][batch-in-block][slice row]
                dist.all_gather_into_tensor(flat, mine.contiguous())
                q_blk[:count].view(count, world, sl, d).copy_(flat.view(world, count, sl, d).permute(1, 0, 2, 3))   # -> each batch's nq vectors in order
            else:                                                                           # ragged slices: batch by batch
                for s_i in range(count):
                    mine = encoder.forward(tok_ids[q_lo:q_hi], tok_mask[q_lo:q_hi], 0)
                    dist.all_gather(list(q_blk[s_i].split(counts)), mine)
        smallest = min(max(0, min(per, total - r * per)) for r in range(world))      # rows of the smallest shard: known to every rank
        can_defer = k <= smallest and world * k <= searcher.DEVICE_MERGE_MAX         # the deferred / blocking decision must be the same on all ranks
        res = []
        for s_i in range(count):
            qs = q_blk[s_i] if encoder is not None else q_vec
            res.append(searcher.search_deferred(qs, k) if can_defer else searcher.search(qs, k))
        if can_defer:
            searcher.finish_deferred()                                    # ONE host synchronisation per block (+ the certificate agreement)
        return res

    def step():
        if encoder is None:
            qv = q_vec
        elif world == 1:
            qv = encoder.forward(tok_ids, tok_mask, 0)                  # [nq, d] fp32 on the device
        else:
            mine = encoder.forward(tok_ids[q_lo:q_hi], tok_mask[q_lo:q_hi], 0)
            if len(set(counts)) == 1:
                dist.all_gather_into_tensor(q_all, mine)
            else:
                dist.all_gather(list(q_all.split(counts)), mine)
            qv = q_all
        return searcher.search(qv, k)

    use_async = world == 1 and not args.sync_search
    use_blocks = world > 1 and args.encode_split == "batch"
    if use_blocks:
        for b0 in range(0, args.warmup, world):
            block_of_steps(b0, min(world, args.warmup - b0))
    else:
        for i in range(args.warmup):
            step_async(i) if use_async else step()
    drain()
    torch.cuda.synchronize()
    index.stats(reset=True)
    coarse_ms = []
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    coarse_log.clear()
    if use_blocks:
        for b0 in range(0, args.steps, world):
            block_of_steps(b0, min(world, args.steps - b0))
            coarse_ms.append(index.stats()["last_coarse_ms"])
    for i in range(0 if use_blocks else args.steps):
        if use_async:
            step_async(i)
        else:
            step()
            coarse_ms.append(index.stats()["last_coarse_ms"])
    if use_async:
        drain()
        coarse_ms = list(coarse_log)
        assert len(coarse_ms) == args.steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    enc_info = None
    if encoder is not None:
        # passages-encoded/s leg of the metric (BASELINE config 1/2 passage shape: 128 tokens), same protocol
        for _ in range(max(1, args.warmup)):
            encoder.forward(pas_ids, pas_mask, 0)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            encoder.forward(pas_ids, pas_mask, 0)
        torch.cuda.synchronize()
        te = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
        te = float(te.item())
        fl = BS.encoder_flops(encoder.cfg, pas_mask.sum(1))
        enc_info = {"operand_dtype": encoder.operand_dtype, "residual_lo": encoder.residual_lo, "passages_per_s": args.passages * world * args.steps / te, "batch": args.passages, "tokens": args.passage_tokens,
                    "tflops_per_gpu": fl * args.steps / te / 1e12, "frac_of_mfma_peak": fl * args.steps / te / PEAK_MFMA_DENSE_16BIT,
                    "algorithmic_gflop_per_passage": fl / args.passages / 1e9}
        # same batch with ragged lengths (clip(N(0.86 S, 0.2 S), 16, S)): the encoder packs attended tokens, padding costs no FLOPs
        rag_ids, rag_mask = BS.synthetic_tokens(dev, args.passages, args.passage_tokens, seed=1, ragged=True)
        encoder.forward(rag_ids, rag_mask, 0)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            encoder.forward(rag_ids, rag_mask, 0)
        torch.cuda.synchronize()
        tr = time.perf_counter() - t2
        flr = BS.encoder_flops(encoder.cfg, rag_mask.sum(1))
        enc_info["ragged"] = {"passages_per_s_per_gpu": args.passages * args.steps / tr, "mean_tokens": float(rag_mask.sum(1).float().mean().item()),
                              "tflops_per_gpu": flr * args.steps / tr / 1e12}

    if rank == 0:
        st = index.stats()
        # HBM traffic of the dominant kernel: PMC counters cannot be collected from inside this process; the figure comes from the
        # committed rocprofv3 --pmc passes of this same workload (tools/profile_round.sh -> profiles/rNN/traffic.json), newest round.
        traffic, traffic_src, traffic_range = None, None, None
        import glob
        for cand in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "traffic*.json")), reverse=True):
            with open(cand) as f:
                tj = json.load(f)
            if (tj.get("rows"), tj.get("dim"), tj.get("queries"), tj.get("topk"), tj.get("coarse_dtype")) == (n, d, nq, k, args.coarse_dtype):
                traffic = tj["hbm_bytes_per_scan"] / 1e9
                traffic_src = os.path.relpath(cand, REPO)
                traffic_range = [v / 1e9 for v in tj["hbm_bytes_per_scan_range"]] if tj.get("hbm_bytes_per_scan_range") else None
                break
        # MFMA-pipe utilisation of the same kernel from the committed counter pass (tools/profile_round.sh: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES
        # GRBM_GUI_ACTIVE; tools/pmc_mfma.py): busy cycles of the matrix pipes / (kernel cycles x 1024 SIMDs)
        mfma_busy, mfma_src = None, None
        for cand in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "mfma_busy.json")), reverse=True) if traffic is not None else []:   # same-workload rule as the traffic figure
            with open(cand) as f:
                mj = json.load(f)
            ent = [v for kname, v in mj.items() if "k_coarse" in kname and "false, false" in kname and "mfma_busy" in v]
            if ent:
                mfma_busy, mfma_src = ent[0]["mfma_busy"], os.path.relpath(cand, REPO)
                break
        # sustained shader clock of the same kernel from the committed counter pass (tools/pmc_clock.py: GRBM_GUI_ACTIVE cycles per scan / HIP-event time per
        # scan of the same run): the 2.5 PFLOP/s nameplate assumes 2.4 GHz, the chip holds ~1.6 GHz under this load
        clock_ghz, clock_src = None, None
        for cand in sorted(glob.glob(os.path.join(REPO, "profiles", "r*", "clock.json")), reverse=True) if traffic is not None else []:
            with open(cand) as f:
                cj = json.load(f)
            if (cj.get("rows"), cj.get("queries")) == (n, nq) and cj.get("sustained_clock_ghz"):
                clock_ghz, clock_src = cj["sustained_clock_ghz"], os.path.relpath(cand, REPO)
                break
        ms_step = dt / args.steps * 1e3
        coarse = float(np.mean(coarse_ms)) * 1e-3           # seconds per coarse scan (sum of its round launches)
        flops = 2.0 * nq * n * d                             # algorithmic: every query against every row of the shard
        out = {
            "metric": f"queries/sec ({'encode + ' if encoder is not None else 'search only: '}exact top-{k} search), e5-large-v2 shape, {total / 1e6:g}M x {d} corpus",
            "value": nq * args.steps / dt, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": (f"encoder: {encoder.operand_dtype} MFMA operands, fp32 accumulate" + (" + residual low half" if encoder.residual_lo else "") + "; search: "
                      if encoder is not None else "") + args.coarse_dtype + " MFMA coarse scan / fp64 exact re-rank",
            "data": "synthetic",
            "config": {"workload": f"BASELINE metric config: {total}x{d} {args.coarse_dtype}+fp32 corpus ({args.corpus_dist}) resident in HBM (row-sharded over "
                                   f"{world} GPU(s)), {nq}-query batch ({args.query_tokens} tokens) encoded then searched, brute-force top-{k}",
                       "rows_per_gpu": n, "dim": d, "queries": nq, "topk": k, "total_rows": total, "corpus_dist": args.corpus_dist,
                       "encoder_in_step": encoder is not None,
                       "collective": (searcher.collective if world > 1 else None),
                       "deferred_batches_re_exchanged": searcher.redone,
                       "parallelism": (f"corpus row-sharded x{world}; blocks of up to {world} steps: every rank encodes its 1/{world} slice of each batch of the block "
                                       f"in one forward (one all-gather of the block's query vectors), then per batch: enqueue-only local search, all-gather of "
                                       f"per-shard top-k, device merge; one host synchronisation per block"
                                       if use_blocks else
                                       f"corpus row-sharded x{world}, query batch split x{world} for encoding, all-gather of query vectors and of "
                                       f"per-shard top-k, device merge")},
            "roofline": {"bound": "mfma", "achieved": flops / coarse / 1e12, "peak": PEAK_MFMA_DENSE_16BIT / 1e12, "unit": "TFLOP/s",
                         "frac": flops / coarse / PEAK_MFMA_DENSE_16BIT, "traffic": traffic, "traffic_unit": "GB per scan (FETCH_SIZE x2 + WRITE_SIZE)",
                         "traffic_source": traffic_src, "traffic_range_over_boxes": traffic_range, "mfma_busy": mfma_busy, "mfma_busy_source": mfma_src,
                         "algorithmic_gb": n * d * 2 / 1e9, "kernel": "k_coarse", "rows_scanned": n,
                         "launch_ms": coarse * 1e3,
                         "sustained_clock_ghz": clock_ghz, "sustained_clock_source": clock_src,
                         "frac_of_clock_adjusted_peak": (flops / coarse / PEAK_MFMA_DENSE_16BIT * 2.4 / clock_ghz) if clock_ghz else None,
                         "note": "one 'launch' = one coarse scan of the shard = the sum of its k_coarse round launches (4 at 5M rows), HIP events "
                                 "around each launch on its stream; algorithmic FLOPs = 2*nq*rows*dim; the bf16 MFMA-only loop measured on this "
                                 "device sustains ~1.6-1.7 PFLOP/s on random data (tools/gemm_bench.hip); peak = the 2.4-GHz nameplate, the kernel's counter-measured "
                                 "sustained clock is in sustained_clock_ghz: frac_of_clock_adjusted_peak = frac * 2.4 / clock = what the chip offers at the clock it holds"},
            "encode": enc_info,
            "search_stats": {kk: st[kk] for kk in ("queries", "certified", "fallback", "fine", "exact", "overflow", "reranked_rows", "coarse_rounds", "fine_rounds", "marked_passes", "marked_rows")},
        }
        if world == 1 and encoder is not None and not args.no_latency:
            out["latency"] = latency_block(args, encoder, index, dev)
        if world == 1 and encoder is not None and not args.no_entry_point:
            out["encode"]["entry_point"] = entry_point_block(args, encoder, dev)
        if world == 1 and not args.no_surface:
            out["surface"] = surface_block(args, index, q_vec, dev)
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = cpu_baseline(args, q_vec.cpu().numpy(), encoder is not None, args.corpus_dist, index)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
