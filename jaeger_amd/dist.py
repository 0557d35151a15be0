"""Multi-GPU layer: contig sharding + the final gather.

Windows are independent at inference (BatchNorm / NMD use moving statistics,
``nnlib/v2/layers.py:918-920``, ``nnlib/v2/nmd.py:73-74``), so whole contigs are
dealt to ranks (one process per GPU) and the only exchange is one gather of the
per-window outputs to rank 0 (RCCL over xGMI with backend "nccl"; "gloo" in the
CPU tests).  The reference runs ``predict`` on a single visible GPU
(``commands/predict.py:602``) - there is no collective pattern to mirror.
"""

from __future__ import annotations

import numpy as np


def lpt_partition(weights, n_parts: int) -> list[np.ndarray]:
    """Greedy longest-processing-time partition of items (contigs weighted by their
    window count) into ``n_parts`` groups; each group keeps the original order so a
    contig's windows stay contiguous (``pred_to_dict`` splits on the is-last flag,
    ``postprocess/collect.py:259-293``)."""
    w = np.asarray(weights, np.int64)
    order = np.argsort(-w, kind="stable")
    loads = np.zeros(n_parts, np.int64)
    owner = np.empty(w.size, np.int64)
    for i in order:
        p = int(np.argmin(loads))
        owner[i] = p
        loads[p] += w[i]
    return [np.nonzero(owner == p)[0] for p in range(n_parts)]


def gather_rows(local, dst: int = 0):
    """Gather variable-length row blocks (torch tensors, dim 0 = windows) to ``dst``.

    Returns the list of per-rank tensors on ``dst`` (``None`` elsewhere).  One size
    exchange + one padded gather: the payload is (windows x outputs) f32 - tens of MB
    at most - so the collective is latency-bound on xGMI."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    n_max = max(sizes) if sizes else 0
    padded = local
    if local.shape[0] != n_max:
        padded = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return [b[:s] for b, s in zip(bufs, sizes)]


def restore_order(parts: list[np.ndarray], groups: list[np.ndarray], rows_per_item: np.ndarray) -> np.ndarray:
    """Undo :func:`lpt_partition`: ``parts[p]`` holds the rows of the items ``groups[p]``
    (in that order, ``rows_per_item[i]`` rows each); returns rows in item order."""
    rows_per_item = np.asarray(rows_per_item, np.int64)
    total = int(rows_per_item.sum())
    first = np.cumsum(rows_per_item) - rows_per_item
    tail = parts[0].shape[1:] if parts else ()
    out = np.zeros((total,) + tuple(tail), parts[0].dtype if parts else np.float32)
    for part, items in zip(parts, groups):
        pos = 0
        for i in items:
            n = int(rows_per_item[i])
            out[first[i]:first[i] + n] = part[pos:pos + n]
            pos += n
    return out
