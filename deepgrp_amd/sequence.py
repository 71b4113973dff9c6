"""deepgrp_amd.sequence -- mirror of the reference's Cython module deepgrp/sequence.pyx
(stub: deepgrp/sequence.pyi) on top of the HIP kernels.  Same function names, argument meaning,
return types and error behaviour; arrays cross PCIe on every call, so the CLI uses the fused
device pipeline (deepgrp_amd.pipeline) instead and these stay for API parity."""
from __future__ import annotations

import ctypes as C
from typing import Iterator, Tuple

import numpy as np
import torch

from ._lib import check, lib
from .pipeline import ContigPipeline, require_gpu, stream_ptr


def one_hot_encode_dna_sequence(sequence: str) -> Tuple[int, np.ndarray]:
    """One hot encodes sequence, drops leading and trailing N's (sequence.pyx:55-58):
    returns (startpos, int8 [5, N]).  An all-N sequence raises ValueError like np.zeros does in
    the reference."""
    raw = sequence.encode("utf-8")
    dev = require_gpu()
    st, kept = C.c_int64(0), C.c_int64(0)
    host = np.frombuffer(raw, dtype=np.uint8)
    check(lib().dgrp_strip_n(host.ctypes.data_as(C.c_void_p) if len(raw) else None, len(raw), C.byref(st), C.byref(kept)))
    if kept.value < 0:
        raise ValueError("negative dimensions are not allowed")
    n = kept.value
    if n == 0:
        return st.value, np.zeros((5, 0), np.int8)
    d_seq = torch.from_numpy(host[st.value:st.value + n].copy()).to(dev)
    d_out = torch.empty((5, n), dtype=torch.int8, device=dev)
    check(lib().dgrp_onehot(d_seq.data_ptr(), n, d_out.data_ptr(), stream_ptr()), "dgrp_onehot")
    return st.value, d_out.cpu().numpy()


def get_max(output: np.ndarray, inputs: np.ndarray, stride: int) -> np.ndarray:
    """In-place overlap max-merge (sequence.pyx:65-76 -> maxcalc.c:10-24).  `output` float32
    [rows, C] C-contiguous, `inputs` float32 [b, T, C] C-contiguous.  Like the Cython signature,
    a non-ndarray argument raises TypeError (prediction.predict relies on that) and a wrong dtype /
    ndim / layout raises ValueError.  Unlike the reference, writes past `rows` are clipped."""
    for name, a, nd in (("output", output, 2), ("inputs", inputs, 3)):
        if isinstance(a, torch.Tensor) and a.is_cuda:
            continue
        if not isinstance(a, np.ndarray):
            raise TypeError(f"Argument '{name}' has incorrect type (expected numpy.ndarray, got {type(a).__name__})")
        if a.dtype != np.float32:
            raise ValueError(f"Buffer dtype mismatch, expected 'float32_t' but got '{a.dtype}'")
        if a.ndim != nd:
            raise ValueError(f"Buffer has wrong number of dimensions (expected {nd}, got {a.ndim})")
        if not a.flags.c_contiguous:
            raise ValueError("ndarray is not C-contiguous")
    dev = require_gpu()
    on_device = isinstance(output, torch.Tensor)
    d_out = output if on_device else torch.from_numpy(output).to(dev)
    d_in = inputs if isinstance(inputs, torch.Tensor) else torch.from_numpy(inputs).to(dev)
    b, d0, d1 = d_in.shape
    if d_out.shape[1] != d1 and d_out.numel():
        pass   # the reference never checks; flat indexing is what matters
    check(lib().dgrp_get_max(d_out.data_ptr(), d_out.numel() // max(d1, 1), d_in.data_ptr(), d0, d1, int(stride), b,
                             stream_ptr()), "dgrp_get_max")
    if not on_device:
        output[...] = d_out.cpu().numpy()
    return output


def get_segments(classes: np.ndarray, startpos: int) -> Tuple[int, int, int]:
    """Start, end and label of the next non-null segment at or after `startpos`
    (sequence.pyx:38-53) incl. its `length = size - 1` behaviour.  Scalar host logic."""
    classes = np.asarray(classes)
    if classes.ndim != 1:
        raise ValueError(f"Buffer has wrong number of dimensions (expected 1, got {classes.ndim})")
    if classes.dtype != np.int64:
        raise ValueError(f"Buffer dtype mismatch, expected 'long' but got '{classes.dtype}'")
    length = classes.size - 1
    tail = classes[startpos:length]
    nz = np.flatnonzero(tail)
    startpos = startpos + int(nz[0]) if nz.size else max(startpos, length)
    label = int(classes[startpos])
    rest = classes[startpos + 1:length]
    diff = np.flatnonzero(rest != label)
    end = startpos + 1 + (int(diff[0]) if diff.size else max(rest.size, 0))
    return startpos, end, label


def yield_segments(classes: np.ndarray, start_offset: int) -> Iterator[Tuple[int, int, int]]:
    """Iterator over continuous segments (sequence.pyx:79-85): every run of equal non-zero labels
    in classes[:-1] plus the last element as a segment of its own (whatever its label).
    Segments are extracted on the GPU in one pass (dgrp_segments)."""
    classes = np.asarray(classes)
    if classes.size == 0:
        return
    dev = require_gpu()
    d_lab = torch.from_numpy(classes.astype(np.int8)).to(dev)
    rows = ContigPipeline.segments(None, d_lab, int(start_offset), 0)
    for r in rows:
        yield int(r["start"]), int(r["end"]), int(r["label"])
    if classes[-1] == 0:
        yield classes.size - 1 + start_offset, classes.size + start_offset, 0
