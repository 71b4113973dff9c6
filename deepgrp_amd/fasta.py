"""Fast (multi-)FASTA ingest with the exact semantics of the reference's line loop
(deepgrp/__main__.py:20-43): every line is stripped, '>' lines open a record whose header is the
rest of the line, other lines are upper-cased and concatenated, a record without header is
dropped and a blank line raises IndexError.

The reference loop costs ~1 us per line in Python (seconds for a chromosome); here a record whose
body is plain ("ACGT...\\n" lines, optional CRLF, no other whitespace, ASCII) is assembled with
bytes.translate/upper at memory speed and anything else falls back to the reference loop for that
record, so results -- including where an exception is raised -- are identical.
"""
from __future__ import annotations

import io
import mmap
import threading
import os
from typing import Iterator, List, Optional, TextIO, Tuple, Union

_DELETE = b"\n\r"
_ODD_WHITESPACE = (b" ", b"\t", b"\x0b", b"\x0c", b"\x1c", b"\x1d", b"\x1e", b"\x1f", b"\x85")


class LineLoop:
    """The reference's line loop (deepgrp/__main__.py:28-43) as an object whose state survives between pieces of a file -- the ONE
    copy of that loop in the package.  `feed` takes an iterable of text lines and yields the records they complete; `flush` yields the
    record still open at the end.  `opened` counts the header lines seen (a caller that needs an order key per record reads it)."""

    __slots__ = ("header", "sequence", "opened", "key", "last_key")

    def __init__(self):
        self.header = ""
        self.sequence: List[str] = []
        self.opened = 0
        self.key = self.last_key = None

    def feed(self, lines, tag=None) -> Iterator[Tuple[str, str]]:
        """`tag` (any sortable value, e.g. the byte offset of the piece) goes into the order key of the records opened by these lines:
        `last_key` = (tag, running number of the header) of the record yielded last."""
        for line in lines:
            line = line.strip()
            if line[0] == ">":                          # a blank line raises IndexError here, as in the reference
                if self.header:
                    self.last_key = self.key
                    yield self.header, "".join(self.sequence)
                self.header = line[1:]
                self.sequence = []
                self.key = (tag, self.opened)
                self.opened += 1
            else:
                self.sequence.append(line.upper())

    def flush(self) -> Iterator[Tuple[str, str]]:
        if self.header:
            self.last_key = self.key
            yield self.header, "".join(self.sequence)
        self.header, self.sequence = "", []


def read_multi_fasta_lines(filestream: TextIO) -> Iterator[Tuple[str, str]]:
    """The reference loop, line by line (used for stdin and as the fallback)."""
    loop = LineLoop()
    yield from loop.feed(filestream)
    yield from loop.flush()


def _text_lines(raw: bytes):
    """The bytes of a piece of a file as the lines `open(path, "r")` would give (locale encoding, universal newlines)."""
    return io.TextIOWrapper(io.BytesIO(raw), encoding=None, newline=None)


_UPPER = bytes.maketrans(bytes(range(ord("a"), ord("z") + 1)), bytes(range(ord("A"), ord("Z") + 1)))


def _plain(body) -> bool:
    """True when stripping every line is the same as deleting CR/LF: ASCII, no whitespace or control
    byte other than line ends, CR only as part of CRLF, no blank line."""
    import numpy as np
    arr = np.frombuffer(body, dtype=np.uint8)
    if arr.size == 0:
        return True
    if int(arr.max()) >= 128:
        return False
    n_ctl = int(np.count_nonzero(arr <= 32))
    n_lf, n_cr, n_crlf = body.count(b"\n"), body.count(b"\r"), body.count(b"\r\n")
    if n_cr != n_crlf or n_ctl != n_lf + n_cr:
        return False                                   # lone CR (a line break in text mode) or other whitespace
    if b"\n\n" in body or b"\n\r\n" in body or body[:1] == b"\n" or body[:2] == b"\r\n":
        return False                                   # blank line: let the reference loop raise
    return True


def read_multi_fasta_file(path: Union[str, os.PathLike]) -> Iterator[Tuple[str, Union[str, bytes]]]:
    """Records of a FASTA file as (header, sequence); the sequence comes back as ASCII bytes on
    the fast path (accepted by the device pipeline as is) or as str from the fallback."""
    size = os.path.getsize(path)
    if size == 0:
        return
    with open(path, "rb") as fh, mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) as mm:
        # chunk boundaries: every "\n>" (a header at the start of a line) and the file start
        starts = [0]
        pos = mm.find(b"\n>")
        while pos != -1:
            starts.append(pos + 1)
            pos = mm.find(b"\n>", pos + 1)
        starts.append(size)
        loop = LineLoop()               # state of the reference loop across chunks
        for a, b in zip(starts[:-1], starts[1:]):
            chunk = mm[a:b]
            nl = chunk.find(b"\n")
            head, body = (chunk, b"") if nl == -1 else (chunk[:nl], chunk[nl + 1:])
            if chunk.startswith(b">") and head.isascii() and b"\r" not in head[:-1] and _plain(body):
                # flush whatever the fallback loop still holds, then emit this record directly
                yield from loop.flush()
                header = head.decode("ascii").strip()[1:]
                seq = body.translate(_UPPER, _DELETE)          # strip line ends + upper() in one pass
                if header:
                    # the reference yields a record when the NEXT header (or EOF) arrives; order is the same
                    yield header, seq
                continue
            # fallback: run the reference loop over this chunk, continuing its state
            yield from loop.feed(_text_lines(chunk))
        yield from loop.flush()


class DeviceRecord:
    """A record whose sequence never existed as a Python string: class indices in HBM.
    `d_idx` is the kept part (leading/trailing N dropped), `startpos` the number of leading N.
    `base` is the ingest group's buffer and `offset` the position of the kept part in it (records of one group can go
    to the GPU as a batch); the `d_idx` view is only made when somebody asks for it."""

    __slots__ = ("startpos", "length", "base", "offset", "_view")

    def __init__(self, startpos, d_idx, length, base=None, offset=0):
        self.startpos, self.length = startpos, length
        self.base, self.offset = base, offset
        self._view = d_idx

    @property
    def d_idx(self):
        if self._view is None:
            self._view = self.base[self.offset:self.offset + max(self.length, 0)]
        return self._view


RESIDENT_BYTES = int(os.environ.get("DGRP_FASTA_RESIDENT_BYTES", str(32 << 30)))   # files up to here are uploaded whole (HBM: 288 GB)
# Pinning host memory costs ~2.4 ms per MB once per process, so the staging area stays small (16 MB) and is turned over many
# times: four slabs, each with its own reader thread.  One thread's read(2) from the page cache into pinned memory moves 7.5 GB/s
# -- the bound of a single-reader upload (34 ms per 254 MB) --; preadv releases the GIL, so four readers fill their slabs side by
# side while the copy engine drains the ones already filled.
_SLAB = 4 << 20
_NSLAB = 4
_UPLOAD: dict = {}
_UPLOAD_LOCK = threading.Lock()       # one file at a time through the slabs
UPLOAD_STATS = {"bytes": 0, "uploads": 0}   # file bytes this process has sent to its GPU (the sharded command line reports them per rank)


def _upload_file(path, size: int, dev, offset: int = 0):
    """Bytes [offset, offset + size) of the file in HBM: preadv straight into pinned slabs by one reader thread per slab, each slab
    sent by the copy engine while the others are being filled."""
    import torch
    UPLOAD_STATS["bytes"] += int(size)
    UPLOAD_STATS["uploads"] += 1
    if size <= _SLAB:                     # a small piece: one pageable copy is cheaper than pinning anything
        import numpy as np
        return torch.from_numpy(np.fromfile(path, dtype=np.uint8, count=size, offset=offset)).to(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _UPLOAD:
        from concurrent.futures import ThreadPoolExecutor
        _UPLOAD[key] = ([torch.empty(_SLAB, dtype=torch.uint8, pin_memory=True) for _ in range(_NSLAB)], torch.cuda.Stream(device=dev),
                        ThreadPoolExecutor(max_workers=_NSLAB, thread_name_prefix="dgrp-upload"))
    slabs, copy, pool = _UPLOAD[key]
    with _UPLOAD_LOCK:
        return _upload_through(path, size, dev, slabs, copy, pool, offset)


def _upload_through(path, size: int, dev, slabs, copy, pool, offset: int = 0):
    import torch
    nslab = len(slabs)
    views = [b.numpy() for b in slabs]
    d_file = torch.empty(size, dtype=torch.uint8, device=dev)
    copy.wait_stream(torch.cuda.current_stream())
    fd = os.open(path, os.O_RDONLY)

    def reader(j):
        """Slab j carries the pieces j, j + nslab, ... of the file: fill it, hand it to the copy engine, wait for it to leave."""
        sent = None
        with torch.cuda.stream(copy):                            # (device and stream are per thread)
            for o in range(j * _SLAB, size, nslab * _SLAB):
                want = min(_SLAB, size - o)
                if sent is not None:
                    sent.synchronize()                           # the slab's previous content has left
                got = 0
                while got < want:
                    k = os.preadv(fd, [views[j][got:want]], offset + o + got)
                    if not k:
                        raise OSError(f"{path}: shorter than its size at open ({offset + o + got} of {offset + size} bytes)")
                    got += k
                d_file[o:o + want].copy_(slabs[j][:want], non_blocking=True)
                sent = torch.cuda.Event()
                sent.record(copy)
        if sent is not None:
            sent.synchronize()                                   # the slab is free for the next caller

    try:
        jobs = [pool.submit(reader, j) for j in range(nslab)]
        err = None
        for job in jobs:
            try:
                job.result()
            except BaseException as e:                           # noqa: BLE001 -- every reader is joined before the first error goes up
                err = err or e
        if err is not None:
            raise err
    finally:
        os.close(fd)
    torch.cuda.current_stream().wait_stream(copy)
    return d_file


def _device_chunks(L, d_file, size: int):
    """(chunk starts, first line feed of each chunk or `size`) of the uploaded file: dgrp_fasta_chunks."""
    import ctypes as C

    import numpy as np
    import torch

    from ._lib import check
    from .pipeline import stream_ptr
    cap = 4096
    while True:
        wb = L.dgrp_fasta_chunks_workspace_bytes(cap)
        work = torch.empty(wb, dtype=torch.uint8, device=d_file.device)
        st, lf, n = np.empty(cap, np.int64), np.empty(cap, np.int64), C.c_int64()
        check(L.dgrp_fasta_chunks(d_file.data_ptr(), size, cap, st.ctypes.data, lf.ctypes.data, C.byref(n), work.data_ptr(), wb,
                                  stream_ptr()), "dgrp_fasta_chunks")
        if n.value <= cap:
            return st[:n.value], lf[:n.value]
        cap = n.value


def chunk_starts_host(path: Union[str, os.PathLike], lo: int = 0, hi: Optional[int] = None):
    """Byte offsets in [lo, hi) at which a chunk of the file starts -- the file start and every '>' that follows a line feed: the
    pieces the reference loop's state does not cross (every one but possibly the first opens with a header line).  A host scan
    (memchr speed) of that slice only: the ranks of a sharded run take a slice each and share the lists."""
    import numpy as np
    size = os.path.getsize(path)
    hi = size if hi is None else min(hi, size)
    out = [0] if lo == 0 and hi > 0 else []
    if hi - lo <= 0:
        return np.asarray(out, np.int64)
    with open(path, "rb") as fh, mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) as mm:
        pos = mm.find(b"\n>", max(lo - 1, 0), hi)
        while pos != -1:
            out.append(pos + 1)
            pos = mm.find(b"\n>", pos + 1, hi)
    return np.asarray(out, np.int64)


def read_multi_fasta_device(path: Union[str, os.PathLike], group_bytes: int = 256 << 20, group_records: int = 4096):
    """Like read_multi_fasta_file, but plain record bodies are uploaded as raw file bytes and turned
    into class indices on the GPU (dgrp_fasta_encode_batch: line-end removal, upper-casing, N stripping and
    the class lookup of deepgrp/sequence.pyx in one pass).  Yields (header, DeviceRecord) for those
    and (header, str) for records that need the reference loop.

    Records are taken in groups (up to `group_bytes` of file or `group_records` records): one upload, the encode
    kernels of all of them queued back to back, one read-back -- a file of thousands of short records pays one
    wait per group, not one per record."""
    for _key, header, rec in ingest_ranges(path, None, group_bytes, group_records):
        yield header, rec


def ingest_ranges(path: Union[str, os.PathLike], ranges=None, group_bytes: int = 256 << 20, group_records: int = 4096):
    """The records of the byte ranges `ranges` ([(a, b), ...], each starting at a chunk start and ending at one or at the end of the
    file; None = the whole file) as (key, header, record): only those bytes are read and uploaded, so the ranks of a sharded run each
    pay for their own share (deepgrp_amd/__main__.py).  `key` = (byte offset of the record's chunk, number of the header inside the
    piece) sorts the records of a file in file order whichever process produced them."""
    import numpy as np

    from ._lib import lib
    from .pipeline import require_gpu

    dev = require_gpu()
    L = lib()
    size = os.path.getsize(path)
    if size == 0:
        return
    if ranges is None:
        ranges = [(0, size)]
    # ACCESS_COPY: a private, writable mapping (never written) so that torch accepts views of it
    with open(path, "rb") as fh, mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_COPY) as mm:
        for a, b in ranges:
            if not (0 <= a < b <= size) or (a and not (mm[a] == 62 and mm[a - 1] == 10)):
                raise ValueError(f"{path}: [{a}, {b}) is not a range of whole chunks")
            # (no numpy view of the mapping outlives a statement in there: an exception of the reference loop -- the blank line's
            # IndexError -- must find the mapping free of exported buffers when it unwinds through this `with`)
            yield from _ingest_range(L, dev, path, mm, a, b, group_bytes, group_records)


def _ingest_range(L, dev, path, mm, r0: int, r1: int, group_bytes: int, group_records: int):
    """ingest_ranges for one range [r0, r1) of the mapped file; offsets below are relative to r0 unless they say `abs`."""
    import numpy as np
    import torch

    from ._lib import check
    from .pipeline import stream_ptr

    size = r1 - r0
    loop = LineLoop()
    d_file = None
    if size <= RESIDENT_BYTES:
        # the whole range goes up once (pinned slabs, the read of slab k+1 overlaps the DMA of slab k) and the chunk
        # table comes from a device pass over it (dgrp_fasta_chunks): the host touches the header lines only
        d_file = _upload_file(path, size, dev, r0)
        st, first_lf = _device_chunks(L, d_file, size)
        starts_np = np.concatenate([st, [size]]).astype(np.int64)
        gt_np = np.ones(st.size, bool)
        gt_np[0] = mm[r0] == 62
    else:
        # chunk table in numpy (a file of 100 000 records would otherwise spend its time in per-record find()s):
        # chunk starts = range start and every '>' that follows a line feed; end of each chunk's first line
        blk = 1 << 28                                                  # bounded temporaries on multi-GB files
        part = np.frombuffer(mm, dtype=np.uint8, count=size, offset=r0)
        lf_pos = np.concatenate([np.flatnonzero(part[o:o + blk] == 10) + o for o in range(0, size, blk)] or [np.zeros(0, np.int64)])
        nxt = lf_pos + 1
        nxt = nxt[nxt < size]
        starts_np = np.concatenate([[0], nxt[part[nxt] == 62], [size]]).astype(np.int64)
        del nxt
        k = np.searchsorted(lf_pos, starts_np[:-1])                   # first line feed at or after the chunk start
        first_lf = np.where(k < lf_pos.size, lf_pos[np.minimum(k, max(lf_pos.size - 1, 0))] if lf_pos.size else size, size)
        gt_np = part[starts_np[:-1]] == 62
        del lf_pos, k, part
    head_end_np = np.minimum(first_lf, starts_np[1:])                 # no line feed inside the chunk: header runs to its end
    body0_np = np.where(first_lf < starts_np[1:], first_lf + 1, starts_np[1:])
    del first_lf
    starts = starts_np.tolist()
    head_ends, body0_all, gt_all = head_end_np.tolist(), body0_np.tolist(), gt_np.tolist()
    nchunks = len(starts) - 1
    c0 = 0
    while c0 < nchunks:
        # ---- one group of chunks [c0, c1)
        c1 = c0 + 1
        while c1 < nchunks and c1 - c0 < group_records and starts[c1 + 1] - starts[c0] <= group_bytes:
            c1 += 1
        body0s = body0_all[c0:c1]
        # fast path: starts with '>', ASCII header, and no carriage return inside the header line (text mode
        # would end the line there; one directly before the line feed is just CRLF)
        cand = []
        for c in range(c0, c1):
            head = mm[r0 + starts[c]:r0 + head_ends[c]]
            cand.append(gt_all[c] and head.isascii() and b"\r" not in head[:-1])
        g0, g1 = starts[c0], starts[c1]
        infos = np.zeros((c1 - c0, 4), np.int64)
        d_idx = None
        if any(cand):
            # resident range: a view; else the group's bytes go up now (numpy view of the mmap: no host copy)
            if d_file is not None:
                d_raw = d_file[g0:g1]
            else:
                UPLOAD_STATS["bytes"] += g1 - g0
                UPLOAD_STATS["uploads"] += 1
                d_raw = torch.from_numpy(np.frombuffer(mm, dtype=np.uint8, count=g1 - g0, offset=r0 + g0)).to(dev)
            d_idx = torch.empty(g1 - g0, dtype=torch.uint8, device=dev)
            off = np.array([body0s[i] - g0 for i in range(c1 - c0)], np.int64)
            ln = np.array([(starts[c0 + i + 1] - body0s[i]) if cand[i] else 0 for i in range(c1 - c0)], np.int64)
            wb = L.dgrp_fasta_batch_workspace_bytes(c1 - c0, int(ln.sum()))
            work = torch.empty(wb, dtype=torch.uint8, device=dev)
            check(L.dgrp_fasta_encode_batch(d_raw.data_ptr(), c1 - c0, off.ctypes.data, ln.ctypes.data, d_idx.data_ptr(),
                                            infos.ctypes.data, work.data_ptr(), wb, stream_ptr()), "dgrp_fasta_encode_batch")
            del d_raw, work
        info_rows = infos.tolist()                   # plain ints: numpy scalar indexing per record is slow
        for i, c in enumerate(range(c0, c1)):
            a, b = r0 + starts[c], r0 + starts[c + 1]                  # abs
            if cand[i] and info_rows[i][0] == 1:
                for header, seq in loop.flush():
                    yield loop.last_key, header, seq
                header = mm[a:r0 + head_ends[c]].decode("ascii").strip()[1:]
                if header:
                    st, kept = info_rows[i][2], info_rows[i][3]
                    lo = body0s[i] - g0 + st
                    yield (a, 0), header, DeviceRecord(st, None, kept, d_idx, lo)
                continue
            for header, seq in loop.feed(_text_lines(mm[a:b]), tag=a):
                yield loop.last_key, header, seq
        del d_idx
        c0 = c1
    for header, seq in loop.flush():
        yield loop.last_key, header, seq
