"""The handful of helpers of the reference's ``utils/utils.py`` that sit ON the retrieval path:
device placement of collated batches (:190-209) and the in-batch-negative all-gathers (:129-134, :158-188).
Everything else in that file (JSON/TSV IO, logging, seeding) is out of scope and stays in the reference."""
from __future__ import annotations

from typing import List

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
