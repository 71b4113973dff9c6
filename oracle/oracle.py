"""ctypes front-end of the CPU parity checker (oracle/dgrp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py -- never by anything under deepgrp_amd/.

Besides the thin wrappers it holds

* ``nn_forward_numpy`` -- an independent float64 numpy statement of the Keras
  graph in deepgrp/model.py:293-336 (used to cross-check the C statement and
  torch.nn.GRU against each other; the reference's own TensorFlow numerics are
  not available offline: "parity unpinned"),
* ``predict_contig`` -- the whole reference path __main__.py:46-83 + :288-292
  for one record given a callable that plays ``model.predict_on_batch``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Callable, List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


class Seg(C.Structure):
    _fields_ = [("st", C.c_int32), ("en", C.c_int32), ("sc", C.c_double)]


def _build():
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, stdout=subprocess.DEVNULL)


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        _build()
    L = C.CDLL(path)
    i64, i32, vp, dbl = C.c_int64, C.c_int32, C.c_void_p, C.c_double
    L.orc_strip_n.restype = i64
    L.orc_strip_n.argtypes = [vp, i64, C.POINTER(i64)]
    L.orc_encode_idx.argtypes = [vp, i64, vp]
    L.orc_onehot_int8.argtypes = [vp, i64, vp]
    L.orc_window_count.restype = i64
    L.orc_window_count.argtypes = [i64, i64, i64]
    L.orc_windows_f32.argtypes = [vp, i64, i64, i64, i64, vp]
    L.orc_place_row.restype = i64
    L.orc_place_row.argtypes = [i64, i64, i64, i64]
    L.orc_get_max.argtypes = [vp, vp, i64, i64, i64, i64]
    L.orc_merge_all.argtypes = [vp, i64, vp, i64, i64, i64, i64, i64]
    L.orc_np_logf.restype = C.c_float
    L.orc_np_logf.argtypes = [C.c_float]
    L.orc_np_expf.restype = C.c_float
    L.orc_np_expf.argtypes = [C.c_float]
    L.orc_scores.argtypes = [vp, i64, i64, vp, vp]
    L.orc_softmax_argmax.argtypes = [vp, i64, i64, vp, vp]
    L.orc_mss_find_all.restype = C.POINTER(Seg)
    L.orc_mss_find_all.argtypes = [i32, vp, dbl, dbl, C.POINTER(i32)]
    L.orc_free.argtypes = [vp]
    L.orc_find_mss_labels.restype = i32
    L.orc_find_mss_labels.argtypes = [vp, vp, i32, i32, i32, i32, vp, C.POINTER(C.POINTER(Seg))]
    L.orc_get_segment.argtypes = [vp, i64, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.orc_segments.restype = i64
    L.orc_segments.argtypes = [vp, i64, i64, vp]
    for name in ("orc_nn_forward_f32", "orc_nn_forward_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [vp, i64, i64, i64, C.c_int, C.c_int, C.c_int, C.c_int,
                      vp, vp, vp, vp, vp, vp, vp, C.c_int]
    for name in ("orc_lstm_forward_f32", "orc_lstm_forward_f64"):
        f = getattr(L, name)
        f.restype = C.c_int
        f.argtypes = [vp, i64, i64, i64, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int]
    L.orc_confusion_matrix.restype = i64
    L.orc_confusion_matrix.argtypes = [vp, vp, i64, vp]
    L.orc_filter_segments.restype = None
    L.orc_filter_segments.argtypes = [vp, i64, i64]
    L.orc_max_threads.restype = C.c_int
    _LIB = L
    return L


def ref_c() -> Optional[C.CDLL]:
    """The reference's own mss.c + maxcalc.c compiled by ``make -C oracle ref``
    (oracle/_ref/libdeepgrp_ref_c.so), or None when it was not built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libdeepgrp_ref_c.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.mss_find_all.restype = C.POINTER(Seg)
        R.mss_find_all.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_int)]
        R._get_max.restype = C.c_void_p
        R._get_max.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
        _REF = R
    return _REF


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# --------------------------------------------------------------------------- A2
def strip_n(seq: bytes) -> Tuple[int, int]:
    """(startpos, kept length) as deepgrp/sequence.pyx:27-30; length < 0 for all-N."""
    st = C.c_int64(0)
    buf = np.frombuffer(seq, dtype=np.uint8) if len(seq) else np.zeros(0, np.uint8)
    n = lib().orc_strip_n(_p(buf), len(seq), C.byref(st))
    return st.value, n


def encode_idx(seq: bytes) -> np.ndarray:
    buf = np.frombuffer(seq, dtype=np.uint8)
    out = np.empty(len(seq), np.uint8)
    lib().orc_encode_idx(_p(buf), len(seq), _p(out))
    return out


def one_hot_encode_dna_sequence(sequence: str) -> Tuple[int, np.ndarray]:
    """deepgrp/sequence.pyx:55-58."""
    raw = sequence.encode("utf-8")
    st, n = strip_n(raw)
    if n < 0:
        raise ValueError("negative dimensions are not allowed")
    out = np.zeros((5, n), np.int8)
    if n:
        buf = np.frombuffer(raw, dtype=np.uint8)[st:st + n].copy()
        lib().orc_onehot_int8(_p(buf), n, _p(out))
    return st, out


# --------------------------------------------------------------------------- A3/A5/A6
def window_count(n: int, T: int, s: int) -> int:
    return lib().orc_window_count(n, T, s)


def windows_f32(idx: np.ndarray, T: int, s: int, w0: int, nw: int) -> np.ndarray:
    out = np.empty((nw, T, 5), np.float32)
    lib().orc_windows_f32(_p(np.ascontiguousarray(idx)), T, s, w0, nw, _p(out))
    return out


def place_row(w: int, nwin: int, B: int, s: int) -> int:
    return lib().orc_place_row(w, nwin, B, s)


def get_max(output: np.ndarray, inputs: np.ndarray, stride: int) -> np.ndarray:
    assert output.dtype == np.float32 and inputs.dtype == np.float32
    assert output.flags.c_contiguous and inputs.flags.c_contiguous
    b, d0, d1 = inputs.shape
    lib().orc_get_max(_p(output), _p(inputs), d0, d1, stride, b)
    return output


def merge_all(probs: np.ndarray, N: int, s: int, B: int) -> np.ndarray:
    nwin, T, Cc = probs.shape
    out = np.zeros((N, Cc), np.float32)
    lib().orc_merge_all(_p(out), N, _p(np.ascontiguousarray(probs, np.float32)), nwin, T, Cc, s, B)
    return out


# --------------------------------------------------------------------------- A7/A8
def scores(probs: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    probs = np.ascontiguousarray(probs, np.float32)
    N, Cc = probs.shape
    sc = np.empty(N, np.float64)
    cl = np.empty(N, np.int64)
    lib().orc_scores(_p(probs), N, Cc, _p(sc), _p(cl))
    return sc, cl


def softmax_argmax(a: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    a = np.ascontiguousarray(a, np.float32)
    N, Cc = a.shape
    out = np.empty((N, Cc), np.float32)
    cl = np.empty(N, np.int64)
    lib().orc_softmax_argmax(_p(a), N, Cc, _p(out), _p(cl))
    return out, cl


# --------------------------------------------------------------------------- A9/A10
def mss_find_all(S: np.ndarray, min_sc: float, xdrop: float, use_ref: bool = False):
    S = np.ascontiguousarray(S, np.float64)
    n = C.c_int32(0)
    if use_ref:
        ptr = ref_c().mss_find_all(len(S), _p(S), min_sc, xdrop, C.byref(n))
    else:
        ptr = lib().orc_mss_find_all(len(S), _p(S), min_sc, xdrop, C.byref(n))
    out = [(ptr[i].st, ptr[i].en, ptr[i].sc) for i in range(n.value)]
    if use_ref:
        C.CDLL(None).free(ptr)
    else:
        lib().orc_free(ptr)
    return out


def find_mss_labels(inputs: np.ndarray, label: np.ndarray, nof_labels: int,
                    min_mss_len: int, xdrop_len: int, return_segments: bool = False):
    """Labels (= argmax of the reference's one-hot rows), optionally the segments."""
    inputs = np.ascontiguousarray(inputs, np.float64)
    label = np.ascontiguousarray(label, np.int64)
    n = len(inputs)
    out = np.empty(n, np.int64)
    segp = C.POINTER(Seg)()
    nseg = lib().orc_find_mss_labels(_p(inputs), _p(label), n, nof_labels, min_mss_len, xdrop_len,
                                     _p(out), C.byref(segp))
    segs = [(segp[i].st, segp[i].en, segp[i].sc) for i in range(nseg)]
    lib().orc_free(segp)
    return (out, segs) if return_segments else out


# --------------------------------------------------------------------------- A11
def get_segments(classes: np.ndarray, startpos: int) -> Tuple[int, int, int]:
    classes = np.ascontiguousarray(classes, np.int64)
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    lib().orc_get_segment(_p(classes), classes.size, startpos, C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def segments(classes: np.ndarray, offset: int) -> np.ndarray:
    """int64 [n, 3] rows (start, end, label) with label > 0 (__main__.py:288-292)."""
    classes = np.ascontiguousarray(classes, np.int64)
    n = lib().orc_segments(_p(classes), classes.size, offset, None)
    rec = np.empty((n, 3), np.int64)
    if n:
        lib().orc_segments(_p(classes), classes.size, offset, _p(rec))
    return rec


# --------------------------------------------------------------------------- N2
def confusion_matrix(truelbl: np.ndarray, predictedlbl: np.ndarray) -> np.ndarray:
    """deepgrp.prediction.confusion_matrix (prediction.py:204-222), incl. its raw-label indexing."""
    t = np.ascontiguousarray(truelbl, np.int64).ravel()
    q = np.ascontiguousarray(predictedlbl, np.int64).ravel()
    assert t.size == q.size
    k = lib().orc_confusion_matrix(_p(t), _p(q), t.size, None)
    if k < 0:
        raise ValueError("zero-size array to reduction operation maximum which has no identity")
    cnf = np.zeros((k, k), np.int64)
    if lib().orc_confusion_matrix(_p(t), _p(q), t.size, _p(cnf)) < 0:
        raise IndexError("label out of bounds for the confusion matrix")
    return cnf


def filter_segments(array: np.ndarray, min_len: int = 50) -> np.ndarray:
    """deepgrp.prediction.filter_segments (prediction.py:244-260) on a copy; returns the filtered labels."""
    a = np.ascontiguousarray(array, np.int64).copy()
    lib().orc_filter_segments(_p(a), a.size, int(min_len))
    return a


# --------------------------------------------------------------------------- A4
class Weights:
    """Keras tensors of one DeepGRP model: kernel [5,3u],
    recurrent_kernel [u,3u], bias [2,3u], scale [u] or None, ff_kernel
    [u or 2u, C], ff_bias [C]; gate column order z|r|h."""

    def __init__(self, kernel, recurrent, bias, ff_kernel, ff_bias, scale=None, T=200):
        self.kernel = np.ascontiguousarray(kernel, np.float32)
        self.recurrent = np.ascontiguousarray(recurrent, np.float32)
        self.bias = np.ascontiguousarray(bias, np.float32)
        self.ff_kernel = np.ascontiguousarray(ff_kernel, np.float32)
        self.ff_bias = np.ascontiguousarray(ff_bias, np.float32)
        self.scale = None if scale is None else np.ascontiguousarray(scale, np.float32)
        self.T = int(T)
        self.u = self.recurrent.shape[0]
        self.C = self.ff_bias.shape[0]
        self.attention = self.scale is not None
        assert self.kernel.shape == (5, 3 * self.u)
        assert self.recurrent.shape == (self.u, 3 * self.u)
        assert self.bias.shape == (2, 3 * self.u)
        assert self.ff_kernel.shape == ((2 if self.attention else 1) * self.u, self.C)

    @classmethod
    def random(cls, u, C=5, T=200, attention=False, seed=7, gain=1.0):
        """Initialisers recorded in the reference's tests/test_model.json:
        glorot_uniform kernels, orthogonal recurrent kernel, zero biases,
        glorot_uniform attention scale; optionally scaled by ``gain`` and with
        small random biases so every term of the cell is exercised."""
        rng = np.random.default_rng(seed)

        def glorot(shape):
            lim = np.sqrt(6.0 / (shape[0] + shape[-1]))
            return rng.uniform(-lim, lim, size=shape)

        kernel = glorot((5, 3 * u)) * gain
        a = rng.normal(size=(3 * u, u))
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        recurrent = q.T.copy() * gain          # [u, 3u], orthonormal rows
        bias = rng.normal(scale=0.05, size=(2, 3 * u))
        ffk = glorot(((2 if attention else 1) * u, C)) * gain
        ffb = rng.normal(scale=0.05, size=(C,))
        scale = glorot((u, 1))[:, 0] if attention else None
        return cls(kernel, recurrent, bias, ffk, ffb, scale, T)


def nn_forward(idx: np.ndarray, wts: Weights, s: int, w0: int, nw: int,
               dtype=np.float64, threads: int = 0) -> np.ndarray:
    """probs [nw, T, C] from the C statement (float32 or float64)."""
    L = lib()
    f = L.orc_nn_forward_f64 if dtype == np.float64 else L.orc_nn_forward_f32
    cast = lambda a: np.ascontiguousarray(a, dtype)
    probs = np.empty((nw, wts.T, wts.C), dtype)
    idx = np.ascontiguousarray(idx, np.uint8)
    assert nw == 0 or (w0 + nw - 1) * s + wts.T <= idx.size
    k, r, b, fk, fb = map(cast, (wts.kernel, wts.recurrent, wts.bias, wts.ff_kernel, wts.ff_bias))
    sc = cast(wts.scale) if wts.attention else np.zeros(1, dtype)
    if threads <= 0:
        threads = L.orc_max_threads()
    rc = f(_p(idx), s, w0, nw, wts.T, wts.u, wts.C, int(wts.attention),
           _p(k), _p(r), _p(b), _p(sc), _p(fk), _p(fb), _p(probs), threads)
    if rc != 0:
        raise RuntimeError(f"orc_nn_forward failed: {rc}")
    return probs


class LSTMWeights:
    """Keras tensors of the rnn="LSTM" model (deepgrp/model.py:219-223): kernel [5,4u],
    recurrent_kernel [u,4u], bias [4u] (gate columns i|f|c|o), ff_kernel [u,C], ff_bias [C]."""

    def __init__(self, kernel, recurrent, bias, ff_kernel, ff_bias, T=200):
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        self.kernel, self.recurrent, self.bias = f32(kernel), f32(recurrent), f32(bias).reshape(-1)
        self.ff_kernel, self.ff_bias = f32(ff_kernel), f32(ff_bias)
        self.T = int(T)
        self.u = self.recurrent.shape[0]
        self.C = self.ff_bias.shape[0]
        assert self.kernel.shape == (5, 4 * self.u) and self.recurrent.shape == (self.u, 4 * self.u)
        assert self.bias.shape == (4 * self.u,) and self.ff_kernel.shape == (self.u, self.C)

    @classmethod
    def random(cls, u, C=5, T=200, seed=7, gain=1.0):
        rng = np.random.default_rng(seed)

        def glorot(shape):
            lim = np.sqrt(6.0 / (shape[0] + shape[-1]))
            return rng.uniform(-lim, lim, size=shape)

        q, r = np.linalg.qr(rng.normal(size=(4 * u, u)))
        bias = rng.normal(scale=0.05, size=4 * u)
        bias[u:2 * u] += 1.0                                   # unit_forget_bias=True
        return cls(glorot((5, 4 * u)) * gain, (q * np.sign(np.diag(r))).T * gain, bias, glorot((u, C)) * gain,
                   rng.normal(scale=0.05, size=C), T)


def lstm_forward(idx: np.ndarray, wts: "LSTMWeights", s: int, w0: int, nw: int, dtype=np.float64, threads: int = 0):
    L = lib()
    f = L.orc_lstm_forward_f64 if dtype == np.float64 else L.orc_lstm_forward_f32
    cast = lambda a: np.ascontiguousarray(a, dtype)
    probs = np.empty((nw, wts.T, wts.C), dtype)
    idx = np.ascontiguousarray(idx, np.uint8)
    assert nw == 0 or (w0 + nw - 1) * s + wts.T <= idx.size
    k, r, b, fk, fb = map(cast, (wts.kernel, wts.recurrent, wts.bias, wts.ff_kernel, wts.ff_bias))
    if threads <= 0:
        threads = L.orc_max_threads()
    rc = f(_p(idx), s, w0, nw, wts.T, wts.u, wts.C, _p(k), _p(r), _p(b), _p(fk), _p(fb), _p(probs), threads)
    if rc != 0:
        raise RuntimeError(f"orc_lstm_forward failed: {rc}")
    return probs


_COMP = np.array([3, 2, 1, 0, 4])


def nn_forward_numpy(idx: np.ndarray, wts: Weights, s: int, w0: int, nw: int) -> np.ndarray:
    """Independent float64 numpy statement of model.py:293-336, vectorised over
    windows; written from the layer definitions, not from the C code."""
    T, u = wts.T, wts.u
    W = wts.kernel.astype(np.float64)
    U = wts.recurrent.astype(np.float64)
    bi, br = wts.bias.astype(np.float64)
    starts = (w0 + np.arange(nw)) * s
    win = idx[starts[:, None] + np.arange(T)[None, :]].astype(np.int64)       # [nw, T]
    onehot = np.eye(5)[win]                                                    # Input, model.py:304
    rc = onehot[:, ::-1, :][:, :, _COMP]                                       # ReverseComplement :277-279

    def gru(x):                                                               # [nw, T, 5] -> [nw, T, u]
        h = np.zeros((x.shape[0], u))
        outs = []
        for t in range(T):
            xm = x[:, t, :] @ W + bi
            hm = h @ U + br
            z = 1 / (1 + np.exp(-(xm[:, :u] + hm[:, :u])))
            r = 1 / (1 + np.exp(-(xm[:, u:2 * u] + hm[:, u:2 * u])))
            hh = np.tanh(xm[:, 2 * u:] + r * hm[:, 2 * u:])
            h = z * h + (1 - z) * hh
            outs.append(h)
        return np.stack(outs, axis=1), h

    fwd, hf = gru(onehot)
    rev, hr = gru(rc)
    avg = (fwd + rev) / 2                                                      # :312 / :323
    if wts.attention:
        q = ((hf + hr) / 2)[:, None, :]                                        # :311, :313
        e = (wts.scale.astype(np.float64) * np.tanh(q + avg)).sum(-1)          # AdditiveAttention
        e = e - e.max(axis=1, keepdims=True)
        a = np.exp(e)
        a /= a.sum(axis=1, keepdims=True)
        ctx = (a[:, :, None] * avg).sum(axis=1)                                # [nw, u]
        feat = np.concatenate([np.repeat(ctx[:, None, :], T, axis=1), avg], axis=2)   # :316-319
    else:
        feat = avg
    logits = feat @ wts.ff_kernel.astype(np.float64) + wts.ff_bias.astype(np.float64)
    logits -= logits.max(axis=2, keepdims=True)
    p = np.exp(logits)
    return p / p.sum(axis=2, keepdims=True)


# --------------------------------------------------------------------------- whole path
def predict_merged(idx: np.ndarray, predict_windows: Callable[[int, int], np.ndarray],
                   T: int, C_: int, s: int, B: int) -> np.ndarray:
    """prediction.py:89-111 driven batch by batch exactly as the reference does
    (so the partial-batch placement comes out of the same arithmetic)."""
    N = idx.size
    nwin = window_count(N, T, s)
    out = np.zeros((N, C_), np.float32)
    i = 0
    w = 0
    while w < nwin:
        b = min(B, nwin - w)
        probs = np.ascontiguousarray(predict_windows(w, b), np.float32)
        index = i * b * s
        get_max(out[index:], probs, s)
        w += b
        i += 1
    return out


def labels_from_merged(merged: np.ndarray, min_mss_len: int, xdrop_len: int, use_mss: bool = True):
    if use_mss:
        sc, cl = scores(merged)
        return find_mss_labels(sc, cl, merged.shape[1], min_mss_len, xdrop_len)
    return softmax_argmax(merged)[1]


def predict_contig(seq: str, predict_windows_factory, T: int, C_: int, s: int = 50, B: int = 256,
                   min_mss_len: int = 50, xdrop_len: int = 50, use_mss: bool = True) -> np.ndarray:
    """__main__.py:46-83 + :288-292 for one record: rows (start, end, label)."""
    raw = seq.encode("utf-8")
    st, n = strip_n(raw)
    if n < 0:
        raise ValueError("negative dimensions are not allowed")
    idx = encode_idx(raw[st:st + n])
    merged = predict_merged(idx, predict_windows_factory(idx), T, C_, s, B)
    labels = labels_from_merged(merged, min_mss_len, xdrop_len, use_mss)
    return segments(labels, st)


def read_multi_fasta(lines) -> List[Tuple[str, str]]:
    """__main__.py:20-43 (a blank line raises IndexError there, as here)."""
    out = []
    header = ""
    parts: List[str] = []
    for line in lines:
        line = line.strip()
        if line[0] == ">":
            if header:
                out.append((header, "".join(parts)))
            header = line[1:]
            parts = []
        else:
            parts.append(line.upper())
    if header:
        out.append((header, "".join(parts)))
    return out
