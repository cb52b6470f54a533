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
import threading
import time
from typing import Optional

import torch

from . import feed, tensor_pickle
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
    parser.add_argument("--tokenizer_workers", type=int, default=-1,
                        help="tokenizer worker PROCESSES next to the in-process tokenizer thread (each with its own tokenizer; batches are handed out "
                             "dynamically and come back in order); 0: the thread alone; -1 (default): 0 below 200k passages per rank, else min(4, usable cpus / 4)")
    parser.add_argument("--no_embedding_files", action="store_true",
                        help="do not write corpus_embeddings_*.pkl / passage_id_list_*.pkl (streamed encode straight into a resident index shard)")
    return parser.parse_args(argv)


def shard_range(n: int, rank: int, world: int):
    per = (n + world - 1) // world
    return min(rank * per, n), min((rank + 1) * per, n)


def effective_cpus() -> int:
    """CPUs this process may really use: the affinity mask, capped by the cgroup's CPU quota (a 16-CPU share of a 256-CPU machine shows 256 in both
    ``os.cpu_count()`` and the affinity mask; thread pools sized by those get throttled by the scheduler, and every thread of the group stalls with them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: (t.split()[0], t.split()[1])),):
        try:
            q, period = parse(open(path).read())
            if q != "max":
                n = min(n, max(1, int(int(q) / int(period))))
        except (OSError, ValueError, IndexError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and period > 0:
            n = min(n, max(1, q // period))
    except (OSError, ValueError):
        pass
    return n


WORKERS_FROM_PASSAGES = 200_000


def default_tokenizer_workers(n_passages: int, on_gpu: bool) -> int:
    """``--tokenizer_workers -1``.  The feed always tokenises on one thread of this process, with the tokenizer's own thread pool: on the GPU box (16-CPU share)
    that alone delivers 97 % of the encoder's rate (profiles/r06/feed_probe.txt).  Worker processes add capacity for hosts with a slower tokenizer, at a price:
    each takes 2-3 s to import torch + transformers, during which the CPU burst gets the cgroup throttled (-15 % on a 50 k-passage run), so they are started
    only when this rank's share is large enough to amortise that (>= 200 k passages, ~15 s of encoding): min(4, usable cpus / 4) of them."""
    if not on_gpu or n_passages < WORKERS_FROM_PASSAGES:
        return 0
    return max(0, min(4, effective_cpus() // 4))


class _ShardWriter:
    """The D2H -> ``.pkl`` half of the loop on its own thread: the encode loop hands over (event, pinned rows, passage ids) and goes on launching; this thread waits
    for the copy and appends the rows to the current shard file, which is written as a STREAM (``tensor_pickle.StreamingTensorPickle``: the same tensor pickle the
    reference's ``pickle.dump`` writes, compute_corpus_embeddings.py:101-115, emitted head - rows - tail) straight from the pinned landing buffer: no [n, hidden]
    staging tensor on the host, no ``torch.cat``, and nothing left to do after the last batch.  ``passage_id_list_{s}_{e}.pkl`` follows when ``cap`` rows
    (or the rank's last rows) are complete.  ``stream=False`` (or a torch whose tensor pickle the stream writer does not recognise): rows are buffered in a
    tensor allocated at the file's final size and written with ``pickle.dump``."""

    def __init__(self, folder: str, start: int, end: int, cap: int, bs: int, on_gpu: bool, depth: int = 4, stream: bool = True):
        import queue
        self.folder, self.end, self.cap, self.bs, self.on_gpu = folder, end, cap, bs, on_gpu
        self.stream = bool(stream) and tensor_pickle.available()
        self.file_start, self.rows, self.ids, self.buf, self.out, self.rows_total = start, 0, [], None, None, 0
        self.free: "queue.Queue" = queue.Queue()
        self.work: "queue.Queue" = queue.Queue()
        self.depth, self.made, self.error = depth, 0, None
        self.thread = threading.Thread(target=self._run, daemon=True, name="kirag-amd-shard-writer")
        self.thread.start()

    def host_buffer(self, hidden: int) -> torch.Tensor:
        """A pinned [bs, hidden] landing buffer; blocks while all ``depth`` of them are still waiting to be consumed."""
        if self.made < self.depth and self.free.empty():
            self.made += 1
            return torch.empty((self.bs, hidden), dtype=torch.float32, pin_memory=self.on_gpu)
        while True:
            self.check()
            try:
                return self.free.get(timeout=0.2)
            except Exception:   # queue.Empty
                continue

    def put(self, event, host: torch.Tensor, n_rows: int, ids) -> None:
        self.work.put((event, host, n_rows, ids))

    def check(self) -> None:
        if self.error is not None:
            raise self.error

    def _names(self, upto: int):
        return (os.path.join(self.folder, f"corpus_embeddings_{self.file_start}_{upto - 1}.pkl"),
                os.path.join(self.folder, f"passage_id_list_{self.file_start}_{upto - 1}.pkl"))

    def _finish_file(self) -> None:
        """``self.rows`` rows of the current file are in: close / write the embeddings file, write the id list, start over."""
        if not self.ids:
            return
        upto = self.file_start + self.rows
        emb_file, ids_file = self._names(upto)
        logger.info(f"Finished calculating embeddings from {self.file_start} to {upto - 1}. Saving embeddings to {self.folder} ...")
        if self.out is not None:
            self.out.close()                                 # (raises if fewer rows than announced were streamed: the .tmp file is removed)
        else:
            emb = self.buf if self.rows == self.buf.shape[0] else self.buf[:self.rows].clone()
            with open(emb_file, "wb") as f:
                pickle.dump(emb, f)
        with open(ids_file, "wb") as f:
            pickle.dump(self.ids, f)
        self.file_start, self.rows, self.ids, self.buf, self.out = upto, 0, [], None, None

    def _append(self, emb: torch.Tensor, ids) -> None:
        emb_np = emb.numpy()
        while len(ids):
            if self.buf is None and self.out is None:
                self.rows_total = max(1, min(self.cap, self.end - self.file_start))
                if self.stream:
                    self.out = tensor_pickle.StreamingTensorPickle(self._names(self.file_start + self.rows_total)[0], self.rows_total, emb_np.shape[1])
                else:
                    self.buf = torch.empty((self.rows_total, emb_np.shape[1]), dtype=torch.float32)
                    self.buf_np = self.buf.numpy()
            take = min(len(ids), self.rows_total - self.rows)
            if self.out is not None:
                self.out.append(emb_np[:take])               # file write straight from the pinned landing buffer (the GIL is released in the system call)
            else:
                # numpy, not Tensor.copy_: a 2-MB torch CPU copy fans out over the intra-op pool (128 threads on the GPU box) and the burst gets the whole
                # cgroup throttled - writing the files cost 55 cpu-seconds and a fifth of the encoder's rate that way (profiles/r06/feed_probe.txt)
                self.buf_np[self.rows:self.rows + take] = emb_np[:take]
            self.ids.extend(ids[:take]); self.rows += take
            emb_np, ids = emb_np[take:], ids[take:]
            if self.rows == self.rows_total:
                self._finish_file()

    def _run(self) -> None:
        while True:
            item = self.work.get()
            if item is None:
                return
            event, host, n_rows, ids = item
            try:
                if self.error is None:
                    if event is not None:
                        event.synchronize()
                    self._append(host[:n_rows], list(ids))
            except BaseException as e:   # noqa: BLE001 - re-raised on the encode thread
                self.error = e
            self.free.put(host)

    def close(self, flush: bool = True) -> None:
        self.work.put(None)
        self.thread.join()
        try:
            if flush and self.error is None:
                self._finish_file()                          # a partial last file (only if fewer rows arrived than the rank's range holds)
        except BaseException as e:   # noqa: BLE001
            self.error = self.error or e
        if self.out is not None:                             # an unfinished stream: never leave half a pickle behind
            self.out.abort(); self.out = None
        self.check()


def cal_doc_embeddings(args, model, corpus_dataset, collator, rank: int = 0, world: int = 1, indexer=None, device: Optional[torch.device] = None):
    """Encode this rank's contiguous shard of ``corpus_dataset`` (items ``{"index": int, "passage": str}`` plus
    ``index_to_passage_id``), write the reference's shard files, and (optionally) add the rows to a resident ``indexer``.
    Returns ``(start, end)`` of the rows handled.

    The loop is a three-stage pipeline in which the one Python thread that launches the encoder does nothing else (it used to un-pickle padded
    int64 batches, pin each of them, validate, clone and buffer — a quarter of the encoder's rate was lost here): ``feed.TokenFeed`` puts RAGGED int32 token
    frames into a pinned ring (tokenizer processes; ids validated where they are produced), ``model.doc_packed`` enqueues upload + forward
    (``kr_encoder_forward_packed``: rows bit-identical to the padded forward) and the append to the resident shard, and ``_ShardWriter`` takes the rows off
    the GPU and into the ``.pkl`` files on its own thread."""
    if device is None:
        device = torch.device("cuda:0") if args.local_rank < 0 else torch.device(f"cuda:{args.local_rank}")
    model = model.to(device)
    model.eval()
    folder = os.path.join(args.save_dir, args.name, args.index_folder)
    os.makedirs(folder, exist_ok=True)
    start, end = shard_range(len(corpus_dataset), rank, world)
    bs = max(int(getattr(args, "encode_batch_size", 512)), int(args.per_gpu_batch_size))
    cap = int(args.num_passage_per_index_file)
    write_files = not bool(getattr(args, "no_embedding_files", False))
    depth = int(getattr(args, "prefetch_batches", 2))
    batches = range(start, end, bs)
    on_gpu = torch.device(device).type == "cuda"
    workers = int(getattr(args, "tokenizer_workers", -1))
    if workers < 0:
        workers = default_tokenizer_workers(end - start, on_gpu)
    use_packed = on_gpu and hasattr(model, "doc_packed") and not bool(getattr(args, "padded_feed", False))
    # Token ids are validated on the HOST, before the batch goes anywhere: the HIP forward reports an id outside the vocabulary only after the fact
    # (deferred error word, enc.check() below), by which time the batch's rows would already sit in the resident shard and in the .pkl buffers.
    vocab = getattr(getattr(getattr(model, "encoder", None), "config", None), "vocab_size", None)
    vocab = int(getattr(args, "vocab_size", 0) or 0) or (int(vocab) if vocab else None)

    def make_texts(s):                                       # dataset access: this process (a serve thread of the feed, or the prefetch thread)
        items = [corpus_dataset[i] for i in range(s, min(s + bs, end))]
        return [it["passage"] for it in items], [corpus_dataset.index_to_passage_id[it["index"]] for it in items]

    def collate_checked(s):                                  # host-only models (the CPU tests of the file contract): the collator's own dict, validated here
        texts, ids = make_texts(s)
        enc = collator.encode_doc(texts)
        if vocab is not None and torch.is_tensor(enc.get("input_ids")) and enc["input_ids"].numel():
            r = feed.attended_range(feed.tokens_of(enc)) if torch.is_tensor(enc.get("attention_mask")) and enc["attention_mask"].shape == enc["input_ids"].shape \
                else (int(enc["input_ids"].min()), int(enc["input_ids"].max()))
            if r is not None and (r[0] < 0 or r[1] >= vocab):
                raise ValueError(feed.bad_id_message(ids[0] if ids else "?", vocab, r[0], r[1]))
        return enc, ids

    tokens = None
    if use_packed or workers > 0:
        max_len = int(getattr(collator, "doc_maxlength", 512) or 512)
        tokens = feed.TokenFeed(make_texts, collator, batches, workers, depth, bs, max_len, vocab=vocab, pinned=on_gpu)
        source = tokens
    else:
        source = prefetch_map(collate_checked, batches, depth=depth)
    if indexer is not None and end > start:
        # the rank's row count is known: reserve it once instead of growing the resident shard by copy (each growth of a small index is a hipMalloc + device
        # copy + hipFree that waits for the device: ~20 pipeline stalls while 65 k rows arrive)
        inner = getattr(indexer, "index", None)
        if hasattr(inner, "reserve") and hasattr(inner, "ntotal"):
            inner.reserve(int(inner.ntotal) + (end - start))
    writer = _ShardWriter(folder, start, end, cap, bs, on_gpu, stream=not bool(getattr(args, "buffered_shard_files", False))) if write_files else None
    pad_id = getattr(getattr(collator, "tokenizer", None), "pad_token_id", None) or 0
    ok = False
    t_loop = time.perf_counter()
    try:
        for item in source:
            ev_in = None
            if tokens is not None:
                frame, ids = item, item.doc_ids
                if use_packed and frame.kind == feed.KIND_RAGGED:
                    emb = model.doc_packed(frame.ids, frame.lens, frame.S, frame.T).detach()    # upload from the pinned slot + forward, enqueue-only
                else:
                    cpu_inputs = frame.inputs(pad_id)
                    inputs = {k: (v.pin_memory().to(device, non_blocking=True) if on_gpu else v.to(device)) for k, v in cpu_inputs.items()}
                    emb = model.doc(inputs).detach()
                if on_gpu:
                    ev_in = torch.cuda.Event(); ev_in.record()
                tokens.release(frame, ev_in)
            else:
                cpu_inputs, ids = item
                emb = model.doc(to_device(cpu_inputs, device)).detach()
            if indexer is not None:
                indexer.index_data(ids, emb)                     # device-to-device append into the resident shard (stream-ordered)
            if writer is not None:
                writer.check()
                host = writer.host_buffer(emb.shape[1])
                host[:emb.shape[0]].copy_(emb, non_blocking=on_gpu)
                ev = None
                if on_gpu:
                    ev = torch.cuda.Event(); ev.record()
                writer.put(ev, host, emb.shape[0], ids)
        ok = True
        t_loop = time.perf_counter() - t_loop
    finally:
        t_tail = time.perf_counter()
        if tokens is not None:
            tokens.close()
        if writer is not None:
            if ok:
                writer.close()
            else:
                try:
                    writer.close(flush=False)    # complete files written so far stay; the partial buffer is dropped, as when the reference's loop dies
                except BaseException:   # noqa: BLE001 - the loop's own exception is the one to report
                    pass
    enc = getattr(getattr(model, "encoder", None), "_hip", None)
    if enc is not None:
        enc.check()                                          # token ids outside the vocabulary surface here at the latest
    cal_doc_embeddings.last_feed = {"tokenizer_workers": workers, "packed_forward": bool(use_packed), "loop_s": t_loop, "tail_s": time.perf_counter() - t_tail,
                                    "batches_by_producer": dict(tokens.made_by) if tokens is not None else None}
    return start, end


def pool_map(make_texts, collator, items, workers: int, depth: int):
    """Ordered map over ``workers`` tokenizer PROCESSES, yielding ``({"input_ids", "attention_mask"} int64 [n,S], ids)`` per item — the round 3-5 interface, kept
    as a thin wrapper over ``feed.TokenFeed`` (the frames travel ragged and are re-padded here; ``cal_doc_embeddings`` consumes the frames directly)."""
    items = list(items)
    pad_id = getattr(getattr(collator, "tokenizer", None), "pad_token_id", None) or 0
    tf = feed.TokenFeed(make_texts, collator, items, max(1, workers), depth, 1, 1, local=False)
    for frame in tf:
        yield frame.inputs(pad_id), frame.doc_ids


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
