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


# ---------------------------------------------------------------------------------------------
# One record across several GPUs (SURVEY 8e "if a single contig must be split", N1)
# ---------------------------------------------------------------------------------------------
def window_share(nwin: int, world: int, rank: int, align: int = 16):
    """Contiguous window range [a, b) of `rank`: equal shares in units of `align` windows (the GRU
    kernel's workgroup granularity)."""
    units = (nwin + align - 1) // align
    per, extra = divmod(units, world)
    a = (rank * per + min(rank, extra)) * align
    b = ((rank + 1) * per + min(rank + 1, extra)) * align
    return min(a, nwin), min(b, nwin)


def placement_rows(a: int, b: int, nwin: int, batch: int, step: int, T: int):
    """Row range [lo, hi) the windows [a, b) are merged into, with the reference's partial-last-batch
    offset (deepgrp/prediction.py:104-105)."""
    if b <= a:
        return 0, 0
    nfull, r = divmod(nwin, batch)
    first_short = nfull * batch

    def row(w):
        return w * step if w < first_short else (nfull * r + (w - first_short)) * step

    ends = [a, b - 1]
    if a < first_short < b:
        ends += [first_short - 1, first_short]
    rows = [row(w) for w in ends]
    return min(rows), max(rows) + T


def merged_split(pipe, d_idx: torch.Tensor) -> torch.Tensor:
    """The max-merged probability array [N, C] of ONE record computed by all ranks together: every
    rank runs the fused GRU kernel on its share of the windows into a buffer that covers just its rows,
    rank 0 gathers the slices and max-combines them (neighbouring shares overlap by T - step rows; max
    is exact, so the result is bit-identical to a single-GPU run).  Returned on rank 0 (None elsewhere).
    The sequential post-processing (scores, MSS, segments: ~3 % of the time) then runs on rank 0."""
    from ._lib import check, lib
    from .pipeline import stream_ptr
    L, m = lib(), pipe.model
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = d_idx.device
    n, C_ = d_idx.numel(), m.classes
    nwin = L.dgrp_window_count(n, m.vecsize, pipe.step)
    a, b = window_share(nwin, world, rank)
    lo, hi = placement_rows(a, b, nwin, pipe.batch, pipe.step, m.vecsize)
    local = torch.zeros((max(hi - lo, 1), C_), dtype=torch.float32, device=dev)
    if b > a:
        if m.attention:
            raise NotImplementedError("splitting one record over GPUs is implemented for models without attention")
        m.set_precision(1 if getattr(pipe, "split", False) else 0)
        # absolute row indexing: hand the kernel the address row 0 would have
        base = local.data_ptr() - lo * C_ * 4
        check(L.dgrp_forward_merge(m.handle, d_idx.data_ptr(), n, pipe.step, pipe.batch, a, b - a, base, None, 0, stream_ptr()),
              "dgrp_forward_merge")
    cpu = dist.get_backend() == "gloo"
    cdev = torch.device("cpu") if cpu else dev
    span = torch.tensor([lo, hi], dtype=torch.int64, device=cdev)
    spans = [torch.zeros_like(span) for _ in range(world)]
    dist.all_gather(spans, span)
    spans = [(int(s[0]), int(s[1])) for s in spans]
    cap = max(max(h - l for l, h in spans), 1)
    send = torch.zeros((cap, C_), dtype=torch.float32, device=cdev)
    send[: hi - lo] = local[: hi - lo].to(cdev)
    parts = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, parts, dst=0)
    if rank != 0:
        return None
    out = torch.zeros((n, C_), dtype=torch.float32, device=dev)
    for (l, h), part in zip(spans, parts):
        if h > l:
            torch.maximum(out[l:h], part[: h - l].to(dev), out=out[l:h])
    return out
