"""The handful of helpers of the reference's ``utils/utils.py`` that sit ON the retrieval path:
device placement of collated batches (:190-209) and the in-batch-negative all-gathers (:129-134, :158-188).
Everything else in that file (JSON/TSV IO, logging, seeding) is out of scope and stays in the reference."""
from __future__ import annotations

import queue
import threading
from typing import Callable, Iterable, Iterator, List, TypeVar

import torch
import torch.distributed as dist


def to_device(inputs, device):
    """utils/utils.py:190-209 — dicts (and lists/tuples of dicts / tensors) of tensors to ``device``; anything else raises."""
    def move(d):
        return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in d.items()}
    if isinstance(inputs, dict):
        return move(inputs)
    if isinstance(inputs, (tuple, list)):
        return [move(x) if isinstance(x, dict) else (x.to(device) if torch.is_tensor(x) else x) for x in inputs]
    raise TypeError(f"Currently do not support using <{type(inputs)}> as the type of a batch")


def get_global_tensor_list(local_rank: int, world_size: int, tensor: torch.Tensor) -> List[torch.Tensor]:
    """utils/utils.py:129-134 — all_gather of equally shaped tensors (RCCL over xGMI under backend 'nccl')."""
    if local_rank < 0:
        return [tensor]
    bucket = [torch.zeros_like(tensor) for _ in range(world_size)]
    dist.all_gather(bucket, tensor)
    return bucket


def get_global_embeddings_for_inbatchtraining(local_rank: int, world_size: int, local_embeddings: torch.Tensor) -> torch.Tensor:
    """utils/utils.py:158-174 — concatenate every rank's embeddings, keeping THIS rank's slice attached to autograd."""
    if local_rank < 0:
        return local_embeddings
    gathered = get_global_tensor_list(local_rank, world_size, local_embeddings.detach().clone())
    parts = [local_embeddings if i == local_rank else g.to(local_embeddings.device) for i, g in enumerate(gathered)]
    return torch.cat(parts, dim=0)


def get_global_labels_for_inbatchtraining(local_rank: int, world_size: int, local_labels, local_doc_size: int):
    """utils/utils.py:177-188 — labels of rank i are offset by i * local_doc_size."""
    if local_rank < 0:
        return local_labels
    if local_labels is None:
        return None
    gathered = get_global_tensor_list(local_rank, world_size, local_labels)
    return torch.cat([lab + i * local_doc_size for i, lab in enumerate(gathered)], dim=0).to(local_labels.device)


_T = TypeVar("_T")
_R = TypeVar("_R")


def prefetch_map(fn: Callable[[_T], _R], items: Iterable[_T], depth: int = 2) -> Iterator[_R]:
    """``map(fn, items)`` with ``fn`` running on a background thread, at most ``depth`` results ahead of the consumer, order kept,
    exceptions re-raised at the consumer.  Used to tokenise batch i+1 (HF fast tokenizers release the GIL) while the GPU encodes
    batch i — the reference tokenises inside its single-threaded loop (``compute_corpus_embeddings.py:77-81``, DataLoader
    ``num_workers=0``, ``utils/utils.py:122``), which cannot keep a >10k passages/s encoder fed.
    Length bucketing is not needed on this path: the encoder packs attended tokens on the device, so padding costs no FLOPs."""
    q: "queue.Queue" = queue.Queue(maxsize=max(1, int(depth)))
    stop = threading.Event()
    _END, _ERR = object(), object()

    def put(x) -> bool:
        while not stop.is_set():
            try:
                q.put(x, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            for it in items:
                if not put(fn(it)):
                    return
            put(_END)
        except BaseException as e:   # noqa: BLE001 - forwarded to the consumer
            put((_ERR, e))

    t = threading.Thread(target=work, daemon=True, name="kirag-amd-prefetch")
    t.start()
    try:
        while True:
            x = q.get()
            if x is _END:
                return
            if isinstance(x, tuple) and len(x) == 2 and x[0] is _ERR:
                raise x[1]
            yield x
    finally:
        stop.set()
