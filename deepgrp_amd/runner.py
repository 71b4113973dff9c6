"""How the command line pushes records through the device pipeline: independent records (deepgrp/__main__.py:280-292)
run on a small pool of host threads, one HIP stream each, with ordered results; consecutive short records of one
ingest buffer go to the GPU as one batch (dgrp_predict_batch).  `RecordRunner.results` yields what the
reference's loop would have produced record after record, and raises where that loop would raise."""
from __future__ import annotations

import collections
import os
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import Iterable, Iterator, List, Tuple

import numpy as np
import torch

from .fasta import DeviceRecord
from .pipeline import ContigPipeline

_BATCH = object()             # key slot of a work item that is a batch of records (never equal to a user's key, e.g. a header "batch")
SMALL_RECORD = 1 << 18        # bases: up to here a record may join a batch
BATCH_RECORDS = 4096
BATCH_BYTES = 4 << 30         # workspace a batch may ask for


_STREAMS: dict = {}
_STREAMS_LOCK = threading.Lock()


def _worker_stream(dev: int, i: int):
    """Stream of worker i on device dev, the same object for every runner of the process: torch's allocator caches blocks per
    stream, so a fresh stream per run would leave the previous run's workspaces (14 GB for a 250 Mbp record) reserved for good."""
    with _STREAMS_LOCK:
        pool = _STREAMS.setdefault(dev, [])
        while len(pool) <= i:
            pool.append(torch.cuda.Stream(device=dev))
        return pool[i]


def _format(prefixes: List[bytes], by_contig: bool, rows) -> str:
    """dgrp_format_rows (host code of the library): one pass over the row records, no per-row Python objects."""
    import ctypes as C

    from ._lib import check, lib
    from .pipeline import SEGMENT_DTYPE
    L = lib()
    rows = np.ascontiguousarray(rows, dtype=SEGMENT_DTYPE)
    blob = b"".join(prefixes)
    off = np.zeros(len(prefixes) + 1, np.int64)
    np.cumsum([len(p) for p in prefixes], out=off[1:])
    cap = L.dgrp_format_rows_bound(len(rows), max(len(p) for p in prefixes))
    out = np.empty(cap, np.uint8)
    written = C.c_int64()
    check(L.dgrp_format_rows(blob, off.ctypes.data, len(prefixes), int(by_contig), rows.ctypes.data, len(rows), out.ctypes.data, cap,
                             C.byref(written)), "dgrp_format_rows")
    return out[:written.value].tobytes().decode("utf-8", "surrogateescape")


def rows_text(filename: str, header: str, rows) -> str:
    """The TSV rows of one record (__main__.py:291-292)."""
    if len(rows) == 0:
        return ""
    return _format(["{}\t{}\t".format(filename, header).encode("utf-8", "surrogateescape")], False, rows)


def rows_text_batch(filename: str, headers, rows) -> str:
    """The rows of a batch of records (rows["contig"] = index into `headers`), record order = row order."""
    if len(rows) == 0:
        return ""
    return _format(["{}\t{}\t".format(filename, h).encode("utf-8", "surrogateescape") for h in headers], True, rows)


class RecordRunner:
    """Runs (key, record) pairs -- record = DeviceRecord or sequence text -- and yields results in input order:
    ("one", key, rows) for a record on its own, ("batch", [keys], rows) for a batch (rows["contig"] indexes the keys)."""

    def __init__(self, pipe: ContigPipeline, workers: int = 0, max_bases: int = 1 << 29):
        self.pipe = pipe
        self.workers = workers or int(os.environ.get("DGRP_CLI_WORKERS", "16"))
        self.max_bases = max_bases
        m = pipe.model
        self._T, self._UP = m.vecsize, (m.units + 31) // 32 * 32

    # ---- one record
    def run_record(self, rec, contig: int = 0) -> np.ndarray:
        if isinstance(rec, DeviceRecord):                 # parsed and encoded on the GPU
            if rec.length < 0:
                raise ValueError("negative dimensions are not allowed")     # all-N record, sequence.pyx:32
            return self.pipe.run_idx(rec.d_idx, rec.startpos, contig)
        return self.pipe.run(rec, contig)

    # ---- batching
    def _batch_cost(self, n: int) -> int:
        """Workspace bytes a record of n bases adds to a batch (attention: the avg[t] spill of its windows dominates)."""
        cost = 80 * n
        m = self.pipe.model
        if m.attention:
            cost += len(range(0, n - self._T, self.pipe.step)) * self._T * (self._UP * 4 + m.classes * 4)
        return cost

    def work_items(self, records: Iterable[Tuple[object, object]]) -> Iterator[Tuple[object, object]]:
        """Consecutive short records of one ingest buffer become one (_BATCH, [(key, record), ...]) item, everything
        else stays (key, record)."""
        group: List[Tuple[object, DeviceRecord]] = []
        cost = 0
        batchable = self.pipe.batchable()
        for key, rec in records:
            small = batchable and isinstance(rec, DeviceRecord) and rec.base is not None and 1 <= rec.length <= SMALL_RECORD
            if small:
                c = self._batch_cost(rec.length)
                if group and (group[0][1].base is not rec.base or len(group) >= BATCH_RECORDS or cost + c > BATCH_BYTES):
                    yield _BATCH, group
                    group, cost = [], 0
                group.append((key, rec))
                cost += c
            else:
                if group:
                    yield _BATCH, group
                    group, cost = [], 0
                yield key, rec
        if group:
            yield _BATCH, group

    def run_item(self, item):
        if isinstance(item, list):                        # a batch: rows of all its records, contig = position in the batch
            rows = self.pipe.run_batch(item[0][1].base, [r.offset for _k, r in item], [r.length for _k, r in item],
                                       [r.startpos for _k, r in item], list(range(len(item))))
            return [k for k, _r in item], rows
        return self.run_record(item)

    # ---- ordered execution
    @staticmethod
    def _size(item) -> int:
        if isinstance(item, list):
            return sum(r.length for _k, r in item)
        return item.d_idx.numel() if isinstance(item, DeviceRecord) else len(item)

    def in_order(self, items: Iterable[Tuple[object, object]]) -> Iterator[Tuple[object, object]]:
        """`run_item` for every (key, item) on the pool; yields (key, result) in input order.  While one record is in
        its post-processing (whose fixed-point loop waits on its stream) the next ones are already on the GPU.  The
        bases in flight are bounded; an exception surfaces where the sequential loop would raise it."""
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None
        local = threading.local()
        taken = iter(range(self.workers))

        def task(item):
            if dev is None:
                return self.run_item(item)
            if not hasattr(local, "stream"):
                torch.cuda.set_device(dev)
                local.stream = _worker_stream(dev, next(taken))
            with torch.cuda.stream(local.stream):
                out = self.run_item(item)
                local.stream.synchronize()
            return out

        pending: "collections.deque" = collections.deque()
        inflight = 0
        with ThreadPoolExecutor(max_workers=self.workers) as pool:
            for key, item in items:
                w = self._size(item)
                while pending and (inflight + w > self.max_bases or len(pending) >= 4 * self.workers):
                    k0, f0, w0 = pending.popleft()
                    yield k0, f0.result()
                    inflight -= w0
                pending.append((key, pool.submit(task, item), w))
                inflight += w
                del item
            while pending:
                k0, f0, _w0 = pending.popleft()
                yield k0, f0.result()

    def results(self, records: Iterable[Tuple[object, object]]):
        for key, result in self.in_order(self.work_items(records)):
            if key is _BATCH:
                keys, rows = result
                yield "batch", keys, rows
            else:
                yield "one", key, result
