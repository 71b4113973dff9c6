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


def lpt(weights: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first: item indices per rank (input order inside a rank), deterministic."""
    return shard_contigs(weights, world_size)


def file_chunk_tables(paths: Sequence[str]) -> List[np.ndarray]:
    """Chunk starts (fasta.chunk_starts_host) of every file, on every rank, with each rank scanning 1/world of each file's bytes on
    its host and the lists exchanged (all_gather_object: a few bytes per record)."""
    import os

    from .fasta import chunk_starts_host
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    mine = []
    for path in paths:
        size = os.path.getsize(path)
        mine.append(chunk_starts_host(path, rank * size // world, (rank + 1) * size // world))
    if world == 1:
        return mine
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    return [np.concatenate([parts[r][f] for r in range(world)]) for f in range(len(paths))]


def plan_file_shares(tables: Sequence[np.ndarray], sizes: Sequence[int], extra: Sequence[int], world: int, pieces: int = 8):
    """Which bytes of which file every rank ingests (deepgrp/__main__.py:275-292 of the reference: records are independent).
    `tables[f]` = chunk starts of file f, `sizes[f]` its bytes.  Chunks are work items of their byte length; runs of short chunks
    (files of thousands of contigs) travel together so that a rank's share stays a few contiguous ranges; `extra` are further
    items (records that were parsed on the host: stdin, .npz) by their lengths.  Longest-processing-time-first over all items.
    Returns (per rank [(file, a, b), ...] sorted and coalesced, per rank [extra index, ...])."""
    total = int(sum(sizes)) + int(sum(extra))
    target = max(total // max(world * pieces, 1), 1)
    items: List[Tuple[int, int, int]] = []          # (file, a, b)
    for f, (starts, size) in enumerate(zip(tables, sizes)):
        edges = [int(x) for x in starts] + [int(size)]
        run_a = None
        for a, b in zip(edges[:-1], edges[1:]):
            if b - a >= target:
                if run_a is not None:
                    items.append((f, run_a, a))
                    run_a = None
                items.append((f, a, b))
            else:
                if run_a is None:
                    run_a = a
                if b - run_a >= target:
                    items.append((f, run_a, b))
                    run_a = None
        if run_a is not None:
            items.append((f, run_a, edges[-1]))
    weights = [b - a for _f, a, b in items] + [int(x) for x in extra]
    shares = lpt(weights, world)
    ranges, extras = [], []
    for share in shares:
        mine = sorted(items[i] for i in share if i < len(items))
        merged: List[Tuple[int, int, int]] = []
        for f, a, b in mine:
            if merged and merged[-1][0] == f and merged[-1][2] == a:
                merged[-1] = (f, merged[-1][1], b)
            else:
                merged.append((f, a, b))
        ranges.append(merged)
        extras.append([i - len(items) for i in share if i >= len(items)])
    return ranges, extras


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


def _forward_rows(pipe, d_idx: torch.Tensor, w0: int, nw: int, lo: int, hi: int) -> torch.Tensor:
    """Max-merge of windows [w0, w0 + nw) into a zeroed buffer that stands for rows [lo, hi) of the record's [N, C] array
    (the kernels index rows absolutely: they are handed the address row 0 would have).  Any model: attention models get
    their avg[t] workspace here, chunked like ContigPipeline.merged does."""
    from ._lib import check, lib
    from .pipeline import stream_ptr
    L, m = lib(), pipe.model
    n, C_ = d_idx.numel(), m.classes
    out = torch.zeros((max(hi - lo, 1), C_), dtype=torch.float32, device=d_idx.device)
    if nw <= 0:
        return out
    h = pipe.handle                                # the pipeline's own view of the model (its precision level)
    base = out.data_ptr() - lo * C_ * 4
    chunk = max(16, min(int(getattr(pipe, "chunk_windows", 1 << 20)), L.dgrp_forward_window_chunk(h)))
    work = None
    w = w0
    while w < w0 + nw:
        k = min(chunk, w0 + nw - w)
        wb = L.dgrp_forward_workspace_bytes(h, k)
        if work is None or work.numel() < wb:
            work = torch.empty(max(wb, 256), dtype=torch.uint8, device=d_idx.device)
        check(L.dgrp_forward_merge(h, d_idx.data_ptr(), n, pipe.step, pipe.batch, w, k, base, work.data_ptr(), work.numel(),
                                   stream_ptr()), "dgrp_forward_merge")
        w += k
    return out


def split_plan(n: int, T: int, step: int, batch: int, world: int):
    """Who computes and who owns what when ONE record is spread over `world` ranks (same answer on every rank).
    The full batches' windows [0, first_short) are shared contiguously (16-window units); rank k OWNS the rows from its first
    window's row to the next rank's (the last rank to n), and its windows spill at most T - step rows into the ranges behind it.
    The short last batch (< batch windows) is placed elsewhere by the reference: every rank computes it itself.
    Returns (nwin, first_short, shares [(a, b)], owned [(lo, hi)], short rows (lo, hi))."""
    nwin = len(range(0, n - T, step))
    nfull, r = divmod(nwin, batch)
    first_short = nfull * batch
    shares = [window_share(first_short, world, k) for k in range(world)]
    starts = [min(a * step, n) for a, _b in shares]         # step > T: a trailing rank without windows starts (and ends) at n
    owned = [(0 if k == 0 else starts[k], n if k == world - 1 else starts[k + 1]) for k in range(world)]
    short = (nfull * r * step, min((nfull * r + r - 1) * step + T, n)) if r else (0, 0)
    return nwin, first_short, shares, owned, short


def run_split(pipe, d_idx: torch.Tensor, startpos: int, contig: int = 0) -> np.ndarray:
    """Segment records of ONE record computed by all ranks together (SURVEY 8e last row / 8f N1); returned on rank 0.

    1. every rank runs the forward kernels on its share of the windows into a buffer over the rows it owns plus the
       T - step rows its last windows spill behind them, and on the short last batch (a few windows, placed by Q2);
    2. the spill rows go to the ranks that own them (point to point; max-combining is exact, so the merged rows are
       bit-identical to a single-GPU run) -- (T - step) * C * 4 bytes per rank boundary;
    3. every rank turns its rows into MSS scores and classes (A7); these travel to rank 0 as float32 + int8 = 5 bytes per base
       (the reference's float64 score is a widened float32, prediction.py:53-57) -- instead of 20 bytes of probabilities;
    4. rank 0 runs the sequential part (MSS scan + vote, segments: ~1 % of the time) and returns the rows.
    `-m` (no MSS) gathers the merged rows themselves: its softmax subtracts the record's global maximum."""
    from ._lib import check, lib
    from .pipeline import stream_ptr
    L, m = lib(), pipe.model
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = d_idx.device
    n, C_, T = d_idx.numel(), m.classes, m.vecsize
    cpu = dist.get_backend() == "gloo"
    cdev = torch.device("cpu") if cpu else dev
    nwin, first_short, shares, owned, short = split_plan(n, T, pipe.step, pipe.batch, world)
    a, b = shares[rank]
    own_lo, own_hi = owned[rank]
    spill_hi = min((b - 1) * pipe.step + T, n) if b > a else own_hi
    hi = max(own_hi, spill_hi)
    local = _forward_rows(pipe, d_idx, a, b - a, own_lo, hi)
    # the short last batch: everybody computes it, everybody merges what falls into its own rows
    if short[1] > short[0]:
        sb = _forward_rows(pipe, d_idx, first_short, nwin - first_short, short[0], short[1])
        lo_, hi_ = max(short[0], own_lo), min(short[1], own_hi)
        if hi_ > lo_:
            torch.maximum(local[lo_ - own_lo:hi_ - own_lo], sb[lo_ - short[0]:hi_ - short[0]], out=local[lo_ - own_lo:hi_ - own_lo])
    # spill rows -> their owners (all ranks derive the same list of transfers)
    ops, recvs, keep = [], [], []
    for src in range(world):
        sa, sbw = shares[src]
        s_lo, s_hi = owned[src][1], (min((sbw - 1) * pipe.step + T, n) if sbw > sa else owned[src][1])
        for dst in range(src + 1, world):
            lo_, hi_ = max(s_lo, owned[dst][0]), min(s_hi, owned[dst][1])
            if hi_ <= lo_:
                continue
            if rank == src:
                t = local[lo_ - own_lo:hi_ - own_lo].to(cdev).contiguous()
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, dst))
            elif rank == dst:
                t = torch.empty((hi_ - lo_, C_), dtype=torch.float32, device=cdev)
                recvs.append((lo_, hi_, t))
                ops.append(dist.P2POp(dist.irecv, t, src))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for lo_, hi_, t in recvs:
        torch.maximum(local[lo_ - own_lo:hi_ - own_lo], t.to(dev), out=local[lo_ - own_lo:hi_ - own_lo])
    nown = own_hi - own_lo
    mine = local[:nown]
    counts = [h - l for l, h in owned]
    cap = max(max(counts), 1)
    if pipe.use_mss:
        scores = torch.empty(max(nown, 1), dtype=torch.float64, device=dev)
        cls = torch.empty(max(nown, 1), dtype=torch.int8, device=dev)
        if nown:
            check(L.dgrp_scores(mine.data_ptr(), nown, C_, scores.data_ptr(), cls.data_ptr(), stream_ptr()), "dgrp_scores")
        send_s = torch.zeros(cap, dtype=torch.float32, device=cdev)
        send_c = torch.zeros(cap, dtype=torch.int8, device=cdev)
        send_s[:nown] = scores[:nown].to(torch.float32).to(cdev)          # exact: the score IS a float32 (prediction.py:53-57)
        send_c[:nown] = cls[:nown].to(cdev)
        parts_s = [torch.empty_like(send_s) for _ in range(world)] if rank == 0 else None
        parts_c = [torch.empty_like(send_c) for _ in range(world)] if rank == 0 else None
        dist.gather(send_s, parts_s, dst=0)
        dist.gather(send_c, parts_c, dst=0)
        if rank != 0:
            return np.zeros(0, SEGMENT_DTYPE)
        all_s = torch.cat([p[:c].to(dev) for p, c in zip(parts_s, counts)]).to(torch.float64)
        all_c = torch.cat([p[:c].to(dev) for p, c in zip(parts_c, counts)])
        labels = pipe.labels_from_scores(all_s, all_c)
    else:
        send = torch.zeros((cap, C_), dtype=torch.float32, device=cdev)
        send[:nown] = mine.to(cdev)
        parts = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
        dist.gather(send, parts, dst=0)
        if rank != 0:
            return np.zeros(0, SEGMENT_DTYPE)
        merged = torch.cat([p[:c].to(dev) for p, c in zip(parts, counts)])
        labels = pipe.labels(merged)
    return pipe.segments(labels, startpos, contig)


def raise_together(err: Exception = None) -> None:
    """Collective error check: if any rank holds an exception, EVERY rank raises (its own, or a RuntimeError naming the
    failing rank) instead of the others blocking in the next collective until it times out."""
    if not dist.is_initialized():
        if err is not None:
            raise err
        return
    msgs = [None] * dist.get_world_size()
    dist.all_gather_object(msgs, None if err is None else f"{type(err).__name__}: {err}")
    if err is not None:
        raise err
    bad = [(k, msg) for k, msg in enumerate(msgs) if msg is not None]
    if bad:
        raise RuntimeError(f"rank {bad[0][0]} failed: {bad[0][1]}")
