"""Multi-GPU: one process per GPU, FASTA records sharded by contig, one gather of the
segment records to rank 0.  The reference has no distributed code; records are
independent (deepgrp/__main__.py:280-292 carries no state between them), so there is no
data-path collective -- only the final gather of 24-byte records over RCCL/xGMI."""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .pipeline import SEGMENT_DTYPE


def shard_contigs(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of record indices to ranks; deterministic
    (ties by index).  Returns, per rank, its record indices in input order."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(x) for x in out]


def gather_records(local: np.ndarray, device: torch.device) -> np.ndarray:
    """Concatenate every rank's SEGMENT_DTYPE records on rank 0 (other ranks get an empty
    array), ordered by (contig tag, start): all_gather of the counts, then one padded
    all_gather of the raw bytes -- kilobytes to a few MB, latency-bound on any fabric."""
    if not dist.is_initialized():
        return local                                    # single process: rows are already in record order
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")               # gloo has no GPU all_gather; RCCL ("nccl") takes HBM tensors
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    buf = np.zeros(cap, SEGMENT_DTYPE)
    buf[: local.shape[0]] = local
    mine = torch.from_numpy(buf.view(np.uint8).copy()).to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    if rank != 0:
        return np.zeros(0, SEGMENT_DTYPE)
    rows = [p.cpu().numpy().view(SEGMENT_DTYPE)[:c] for p, c in zip(parts, counts)]
    allrows = np.concatenate(rows) if rows else np.zeros(0, SEGMENT_DTYPE)
    return np.sort(allrows, order=["contig", "start"])


def predict_records_sharded(records: Sequence[Tuple[str, str]], run_one: Callable[[str, int], np.ndarray],
                            device: torch.device) -> np.ndarray:
    """Each rank runs `run_one(sequence, record_index)` on its share; rank 0 returns all rows."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard_contigs([len(s) for _, s in records], world)[rank]
    parts = [run_one(records[i][1], i) for i in mine]
    local = np.concatenate(parts) if parts else np.zeros(0, SEGMENT_DTYPE)
    return gather_records(local, device)
