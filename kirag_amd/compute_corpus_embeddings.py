"""MI355X counterpart of the reference's ``compute_corpus_embeddings.py`` (``cal_doc_embeddings`` :50-134, flags :25-47).

Same flags, same output files — ``corpus_embeddings_{start}_{end}.pkl`` (fp32 ``torch.Tensor [n, hidden]``, rows in corpus-index
order) and ``passage_id_list_{start}_{end}.pkl`` (``List[str]``), ``n <= --num_passage_per_index_file`` — readable by the
reference's ``faiss_index_corpus.py`` and by ``kirag_amd.faiss_index_corpus``.

Different by design: one process per GPU, rank r encodes the CONTIGUOUS corpus rows
``[r*ceil(N/W), (r+1)*ceil(N/W))`` with large packed batches and writes its own shard files — no per-step ``dist.gather`` to
rank 0, no barriers, no growing ``torch.cat`` on the host.  File boundaries therefore follow the rank shards (plus the
per-file row cap) instead of fixed 1M blocks; the concatenation of all files in ``end``-index order is identical.
"""
from __future__ import annotations

import argparse
import logging
import os
import pickle
from typing import Optional

import torch

from .collators import COLLATOR_MAP
from .retriever.retrievers import InBatchRetriever
from .utils import prefetch_map, to_device

logger = logging.getLogger(__file__)


def setup_parser(argv=None):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    parser.add_argument("--local_rank", "--local-rank", type=int, default=-1)
    parser.add_argument("--corpus", type=str, default="2wikimultihopqa", help="the name of the corpus")
    parser.add_argument("--query_maxlength", type=int, default=512, help="the maxinum number of query tokens")
    parser.add_argument("--doc_maxlength", type=int, default=512, help="the maxinum number of doc tokens")
    parser.add_argument("--retriever_name", type=str, default="E5Retriever", choices=["E5Retriever", "BGERetriever"], help="the name of the retriever")
    parser.add_argument("--retriever_model_name_or_path", type=str, default="intfloat/e5-large-v2", help="the name or path of the retriever model")
    parser.add_argument("--tokenizer_name_or_path", type=str, default="intfloat/e5-large-v2", help="the name or path of the tokenizer")
    parser.add_argument("--save_dir", type=str, default="checkpoint")
    parser.add_argument("--name", type=str, default="e5_retriever")
    parser.add_argument("--index_folder", type=str, default="2wikimultihopqa")
    parser.add_argument("--per_gpu_batch_size", type=int, default=8)
    parser.add_argument("--num_passage_per_index_file", type=int, default=1000000)
    # MI355X path: passages per encoder launch (rows are batch-invariant, so this changes speed only)
    parser.add_argument("--encode_batch_size", type=int, default=512)
    parser.add_argument("--prefetch_batches", type=int, default=2, help="batches tokenised ahead of the GPU on a background thread")
    parser.add_argument("--tokenizer_workers", type=int, default=0,
                        help="> 0: tokenise in that many worker PROCESSES (each with its own tokenizer; batches come back in order) instead of "
                             "one background thread - for hosts where one tokenizer cannot keep the encoder fed")
    parser.add_argument("--no_embedding_files", action="store_true",
                        help="do not write corpus_embeddings_*.pkl / passage_id_list_*.pkl (streamed encode straight into a resident index shard)")
    return parser.parse_args(argv)


def shard_range(n: int, rank: int, world: int):
    per = (n + world - 1) // world
    return min(rank * per, n), min((rank + 1) * per, n)


def cal_doc_embeddings(args, model, corpus_dataset, collator, rank: int = 0, world: int = 1, indexer=None, device: Optional[torch.device] = None):
    """Encode this rank's contiguous shard of ``corpus_dataset`` (items ``{"index": int, "passage": str}`` plus
    ``index_to_passage_id``), write the reference's shard files, and (optionally) add the rows to a resident ``indexer``.
    Returns ``(start, end)`` of the rows handled."""
    if device is None:
        device = torch.device("cuda:0") if args.local_rank < 0 else torch.device(f"cuda:{args.local_rank}")
    model = model.to(device)
    model.eval()
    folder = os.path.join(args.save_dir, args.name, args.index_folder)
    os.makedirs(folder, exist_ok=True)
    start, end = shard_range(len(corpus_dataset), rank, world)
    bs = max(int(getattr(args, "encode_batch_size", 512)), int(args.per_gpu_batch_size))
    cap = int(args.num_passage_per_index_file)
    buf, buf_ids, file_start = [], [], start

    def flush(upto):
        nonlocal buf, buf_ids, file_start
        if not buf_ids:
            return
        emb = torch.cat(buf, dim=0)
        logger.info(f"Finished calculating embeddings from {file_start} to {upto - 1}. Saving embeddings to {folder} ...")
        with open(os.path.join(folder, f"corpus_embeddings_{file_start}_{upto - 1}.pkl"), "wb") as f:
            pickle.dump(emb, f)
        with open(os.path.join(folder, f"passage_id_list_{file_start}_{upto - 1}.pkl"), "wb") as f:
            pickle.dump(buf_ids, f)
        buf, buf_ids, file_start = [], [], upto

    def collate(s):                                          # runs on the prefetch thread (or in a worker process): dataset access + tokenisation
        items = [corpus_dataset[i] for i in range(s, min(s + bs, end))]
        return collator.encode_doc([it["passage"] for it in items]), [corpus_dataset.index_to_passage_id[it["index"]] for it in items]

    write_files = not bool(getattr(args, "no_embedding_files", False))
    depth = int(getattr(args, "prefetch_batches", 2))
    workers = int(getattr(args, "tokenizer_workers", 0))
    batches = range(start, end, bs)
    def make_texts(s):
        items = [corpus_dataset[i] for i in range(s, min(s + bs, end))]
        return [it["passage"] for it in items], [corpus_dataset.index_to_passage_id[it["index"]] for it in items]
    source = pool_map(make_texts, collator, batches, workers, depth) if workers > 0 else prefetch_map(collate, batches, depth=depth)
    # The GPU side never waits for the host inside the loop: inputs go up from pinned memory, the forward (device output: asynchronous,
    # kr_encoder_forward) and the append to the resident shard are enqueued on the current stream, the embeddings come down into a small ring of
    # pinned buffers behind an event; the host consumes batch i - 2 (file buffers) while batch i is being encoded.
    ring, pending = [], []

    def drain(keep):
        nonlocal buf, buf_ids, file_start
        while len(pending) > keep:
            ev, host, n_rows, ids = pending.pop(0)
            ev.synchronize()
            emb = host[:n_rows].clone()
            ring.append(host)
            while len(buf_ids) + len(ids) > cap:             # respect the per-file row cap
                take = cap - len(buf_ids)
                buf.append(emb[:take]); buf_ids.extend(ids[:take])
                flush(file_start + cap)
                emb, ids = emb[take:], ids[take:]
            buf.append(emb); buf_ids.extend(ids)
            if len(buf_ids) == cap:
                flush(file_start + cap)

    class _Done:                                             # event stand-in for a host-only model (the CPU tests of the file contract)
        def synchronize(self): pass
    on_gpu = torch.device(device).type == "cuda"
    hidden = None
    # Token ids are validated on the HOST, before the batch goes anywhere: the HIP forward reports an id outside the vocabulary only after the fact
    # (deferred error word, enc.check() below), by which time the batch's rows would already sit in the resident shard and in the .pkl buffers.
    vocab = getattr(getattr(getattr(model, "encoder", None), "config", None), "vocab_size", None)
    vocab = int(getattr(args, "vocab_size", 0) or 0) or (int(vocab) if vocab else None)

    def validate(cpu_inputs, ids):
        t = cpu_inputs.get("input_ids") if isinstance(cpu_inputs, dict) else None
        if vocab is None or not torch.is_tensor(t) or t.is_cuda or t.numel() == 0:
            return
        # ATTENDED positions only: the HIP forward never reads a masked id (k_fill_tokens packs attended positions), and __main__ may add a
        # '[PAD]' token whose id == len(tokenizer) >= the model's vocab_size when the tokenizer has no pad token — a padded batch is not an error
        m = cpu_inputs.get("attention_mask")
        if torch.is_tensor(m) and m.shape == t.shape:
            t = t[m.bool()]
            if t.numel() == 0:
                return
        lo, hi = int(t.min()), int(t.max())
        if lo < 0 or hi >= vocab:
            raise ValueError(f"input_ids of the batch starting at passage id {ids[0] if ids else '?'} contain a token id outside [0, {vocab}) "
                             f"(min {lo}, max {hi}): nothing of this batch was encoded, indexed or written")
    for cpu_inputs, ids in source:
        validate(cpu_inputs, ids)
        if on_gpu:
            inputs = {k: (v.pin_memory().to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in cpu_inputs.items()}
        else:
            inputs = to_device(cpu_inputs, device)
        emb = model.doc(inputs).detach()                     # HIP path (eval mode), stays on the GPU; no host synchronisation
        if indexer is not None:
            indexer.index_data(ids, emb)                     # device-to-device append into the resident shard (stream-ordered)
        if write_files:
            hidden = emb.shape[1]
            host = ring.pop() if ring and ring[-1].shape[0] >= emb.shape[0] else torch.empty((bs, hidden), dtype=torch.float32, pin_memory=on_gpu)
            host[:emb.shape[0]].copy_(emb, non_blocking=on_gpu)
            if on_gpu:
                ev = torch.cuda.Event(); ev.record()
            else:
                ev = _Done()
            pending.append((ev, host, emb.shape[0], ids))
            drain(2)
    drain(0)
    flush(end)
    enc = getattr(getattr(model, "encoder", None), "_hip", None)
    if enc is not None:
        enc.check()                                          # token ids outside the vocabulary surface here at the latest
    return start, end


def pool_map(make_texts, collator, items, workers: int, depth: int):
    """Ordered map over ``workers`` tokenizer PROCESSES (``python -m kirag_amd.tokenize_worker``, started as plain child processes: no fork of a
    process that holds a GPU context and a tokenizer thread pool, no re-import of the caller's ``__main__``).  For every item the parent builds the
    batch's strings (``make_texts(item) -> (texts, ids)``: dataset access stays in the parent), worker ``j % workers`` tokenises batch j with its own
    copy of the collator (one Rust thread each), and the batches are yielded in order, at most ``workers + depth`` ahead of the consumer."""
    import struct
    import subprocess
    import sys
    import threading
    import numpy as np
    items = list(items)
    env = dict(os.environ, TOKENIZERS_PARALLELISM="false", PYTHONPATH=os.pathsep.join(
        [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] + [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep) if p]))
    procs = [subprocess.Popen([sys.executable, "-m", "kirag_amd.tokenize_worker"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
             for _ in range(min(workers, max(1, len(items))))]
    blob = pickle.dumps(collator)

    def send(p, payload):
        p.stdin.write(struct.pack("<Q", len(payload))); p.stdin.write(payload); p.stdin.flush()

    def recv(p):
        head = p.stdout.read(8)
        if len(head) != 8:
            raise RuntimeError("tokenizer worker exited unexpectedly")
        (n,) = struct.unpack("<Q", head)
        return pickle.loads(p.stdout.read(n))
    results, cond, failed = {}, threading.Condition(), []
    window = len(procs) + max(1, depth)
    consumed = [0]

    def serve(w):
        p = procs[w]
        try:
            send(p, blob)
            for j in range(w, len(items), len(procs)):
                with cond:
                    cond.wait_for(lambda: j < consumed[0] + window or failed)
                    if failed:
                        return
                texts, ids = make_texts(items[j])
                send(p, pickle.dumps(texts))
                out = recv(p)
                if isinstance(out, str):
                    raise RuntimeError("tokenizer worker: " + out)
                ii, mm = out
                with cond:
                    results[j] = ({"input_ids": torch.from_numpy(ii.astype(np.int64)), "attention_mask": torch.from_numpy(mm.astype(np.int64))}, ids)
                    cond.notify_all()
        except BaseException as e:   # noqa: BLE001 - forwarded to the consumer
            with cond:
                failed.append(e); cond.notify_all()
    threads = [threading.Thread(target=serve, args=(w,), daemon=True) for w in range(len(procs))]
    for t in threads:
        t.start()
    try:
        for j in range(len(items)):
            with cond:
                cond.wait_for(lambda: j in results or failed)
                if failed:
                    raise failed[0]
                out = results.pop(j)
                consumed[0] = j + 1
                cond.notify_all()
            yield out
    finally:
        with cond:
            failed.append(GeneratorExit()); cond.notify_all()
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()


def main(argv=None):
    import torch.distributed as dist
    from transformers import AutoTokenizer
    args = setup_parser(argv)
    world, rank = 1, 0
    if args.local_rank >= 0 or "RANK" in os.environ:
        if args.local_rank < 0:
            args.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")
        world, rank = dist.get_world_size(), dist.get_rank()
    os.makedirs(os.path.join(args.save_dir, args.name), exist_ok=True)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s: - %(message)s")
    tokenizer = AutoTokenizer.from_pretrained(args.tokenizer_name_or_path)
    if tokenizer.pad_token is None or tokenizer.pad_token_id is None:
        logger.warning("Missing padding token, adding a new pad token!")
        tokenizer.add_special_tokens({"pad_token": '[PAD]'})
    collator = COLLATOR_MAP[args.retriever_name](tokenizer=tokenizer, query_maxlength=args.query_maxlength, doc_maxlength=args.doc_maxlength)
    try:   # corpus datasets are the reference's own (dataset/corpus.py, out of scope): needs the reference on PYTHONPATH
        from utils.const import CORPUS_MAP
    except ImportError as e:
        raise SystemExit("--corpus needs the KiRAG repository on PYTHONPATH (dataset/corpus.py, utils/const.py CORPUS_MAP)") from e
    corpus_dataset = CORPUS_MAP[args.corpus](title_prefix="title: ", passage_prefix="text: ")
    model = InBatchRetriever(retriever_name=args.retriever_name, model_name_or_path=args.retriever_model_name_or_path,
                             local_rank=args.local_rank, temperature=0.01)
    cal_doc_embeddings(args, model, corpus_dataset, collator, rank=rank, world=world)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
