"""Device pipeline for one FASTA record: everything between the sequence bytes and
the segment records stays in HBM.

    bytes --encode--> class idx --GRU/attention + softmax + max-merge--> probs [N,C]
          --scores--> (score f64, class i8) --MSS + vote--> labels i8 --RLE--> records

This is what ``deepgrp predict`` (deepgrp/__main__.py:46-83, :280-292 of the
reference) computes per record; `deepgrp_amd.prediction` / `.sequence` / `.mss`
expose the individual reference functions on top of the same kernels.
PyTorch is used only to own device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch

from ._lib import check, lib

SEGMENT_DTYPE = np.dtype([("start", "<i8"), ("end", "<i8"), ("label", "<i4"), ("contig", "<i4")])


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("deepgrp_amd needs an AMD MI355X (gfx950) visible to PyTorch-ROCm; "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _np_ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class DeviceModel:
    """The tensors ``tf.keras.models.load_model`` yields, packed into MFMA
    fragments in HBM by ``dgrp_model_create``.  Mirrors the two Keras attributes the
    reference reads: ``input_shape`` (__main__.py:270) and ``output_shape`` (:75)."""

    def __init__(self, kernel, recurrent_kernel, bias, ff_kernel, ff_bias, scale=None, vecsize: int = 200,
                 rnn: str = "GRU"):
        require_gpu()
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
        self.kernel, self.recurrent_kernel, self.bias = f32(kernel), f32(recurrent_kernel), f32(bias)
        self.ff_kernel, self.ff_bias = f32(ff_kernel), f32(ff_bias)
        self.scale = None if scale is None else f32(scale).reshape(-1)
        self.units = int(self.recurrent_kernel.shape[0])
        self.classes = int(self.ff_bias.shape[0])
        self.vecsize = int(vecsize)
        self.attention = self.scale is not None
        self.rnn = rnn
        u, c = self.units, self.classes
        h = C.c_void_p()
        if rnn == "LSTM":
            # deepgrp/model.py:219-223: Keras LSTM, gate columns i|f|c|o, one bias vector, never with attention
            self.bias = self.bias.reshape(-1)
            if self.kernel.shape != (5, 4 * u) or self.recurrent_kernel.shape != (u, 4 * u) or self.bias.shape != (4 * u,):
                raise ValueError(f"LSTM tensors have unexpected shapes {self.kernel.shape} {self.recurrent_kernel.shape} {self.bias.shape}")
            if self.attention or self.ff_kernel.shape != (u, c):
                raise ValueError("the LSTM model has no attention and a [units, classes] FF kernel")
            check(lib().dgrp_model_create_lstm(C.byref(h), self.vecsize, u, c, _np_ptr(self.kernel),
                                               _np_ptr(self.recurrent_kernel), _np_ptr(self.bias), _np_ptr(self.ff_kernel),
                                               _np_ptr(self.ff_bias)), "dgrp_model_create_lstm")
            self.handle = h
            self._warn_fp32_path()
            return
        if rnn != "GRU":
            raise ValueError(f"unknown rnn {rnn!r}")
        if self.kernel.shape != (5, 3 * u) or self.recurrent_kernel.shape != (u, 3 * u) or self.bias.shape != (2, 3 * u):
            raise ValueError(f"GRU tensors have unexpected shapes {self.kernel.shape} {self.recurrent_kernel.shape} "
                             f"{self.bias.shape}; expected reset_after GRU with 5 inputs")
        if self.ff_kernel.shape != ((2 if self.attention else 1) * u, c):
            raise ValueError(f"FF kernel shape {self.ff_kernel.shape} does not match units={u}, attention={self.attention}")
        if self.attention and self.scale.shape != (u,):
            raise ValueError("attention scale must have `units` entries")
        check(lib().dgrp_model_create(C.byref(h), self.vecsize, u, c, int(self.attention), _np_ptr(self.kernel),
                                      _np_ptr(self.recurrent_kernel), _np_ptr(self.bias),
                                      _np_ptr(self.scale) if self.attention else None, _np_ptr(self.ff_kernel),
                                      _np_ptr(self.ff_bias)), "dgrp_model_create")
        self.handle = h
        self._warn_fp32_path()

    def _warn_fp32_path(self) -> None:
        """More units than any fused kernel takes: the model runs, on the plain-fp32 kernels (the reference takes any `units`,
        deepgrp/model.py:117,219-229) -- said once, loudly, because it is 10-20x slower than a model of 256 units."""
        self.fp32_only = bool(lib().dgrp_model_flags(self.handle) & 4)
        if self.fp32_only:
            import warnings
            what = f"{self.units} units" if self.units > 256 else f"{self.classes} classes"
            warnings.warn(f"{self.rnn} with {what} is beyond the fused kernels (256 units, 16 classes): every forward pass runs on the "
                          f"plain-fp32 kernels, tens of Mbp/s", RuntimeWarning, stacklevel=3)

    input_shape = property(lambda self: (None, self.vecsize, 5))
    output_shape = property(lambda self: (None, self.vecsize, self.classes))

    # ---- the Keras model surface the reference touches besides predict_on_batch --------------------------------
    def get_config(self) -> dict:
        """``keras.Model.get_config()`` (tests/test_model.py:254-262 of the reference)."""
        from . import model as dgmodel
        if getattr(self, "config", None) is None:
            self.config = dgmodel.keras_config(self.vecsize, self.units, self.classes, self.attention, rnn=self.rnn)
        return self.config["config"]

    def get_weights(self):
        """Tensors in Keras' order: RNN kernel, recurrent kernel, bias, [attention scale], FF kernel, FF bias."""
        scale = [] if self.scale is None else [self.scale.copy()]
        return [self.kernel.copy(), self.recurrent_kernel.copy(), self.bias.copy()] + scale + [self.ff_kernel.copy(), self.ff_bias.copy()]

    def save(self, path: str) -> None:
        """``model.save("*.hdf5")`` (deepgrp/__main__.py:351): the Keras HDF5 layout ``load_model`` reads back."""
        from . import model as dgmodel
        self.get_config()
        dgmodel.save_keras_hdf5(path, self.kernel, self.recurrent_kernel, self.bias, self.ff_kernel, self.ff_bias, self.scale,
                                vecsize=self.vecsize, rnn=self.rnn, config=self.config)

    @property
    def kernel_flags(self) -> int:
        """dgrp_model_flags: bit 0 = the one-reciprocal GRU blend was provably safe for these weights, bit 1 = the
        split-operand kernel is selected."""
        return int(lib().dgrp_model_flags(self.handle))

    @property
    def supports_split(self) -> bool:
        """A split-operand fused kernel (fp16 hi+lo pairs, fp32-grade pre-activations) covers this model (with attention: its
        recurrent pre-pass) -- every model: GRU up to 128 units has the resident-weight kernels, larger GRUs and the LSTM cell
        the streamed one (rnn_stream.hip)."""
        return True

    def set_precision(self, level: int) -> None:
        """dgrp_model_set_precision: 0 = fp16 operands (`--fast`), 1 = split operands (the default of every model), for every later
        call on THIS handle.  Pipelines do not call it: each holds a view of its own (`view`)."""
        check(lib().dgrp_model_set_precision(self.handle, int(level)), "dgrp_model_set_precision")

    def view(self, level: int) -> C.c_void_p:
        """dgrp_model_view: a second handle on the same device buffers with its own precision level (release: dgrp_model_destroy)."""
        v = C.c_void_p()
        check(lib().dgrp_model_view(self.handle, int(level), C.byref(v)), "dgrp_model_view")
        return v

    def close(self):
        if getattr(self, "handle", None):
            lib().dgrp_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- model.predict_on_batch (prediction.py:106) ---------------------------------------
    def forward_windows(self, d_idx: torch.Tensor, step: int, w0: int, nw: int, handle=None) -> torch.Tensor:
        """probs [nw, T, C] (device) of windows w0.. of a class-index tensor; `handle`: a view of this model with a precision
        level of its own (ContigPipeline.handle) instead of the model's."""
        h = handle if handle is not None else self.handle
        probs = torch.empty((nw, self.vecsize, self.classes), dtype=torch.float32, device=d_idx.device)
        wb = lib().dgrp_forward_workspace_bytes(h, nw)
        work = torch.empty(max(wb, 256), dtype=torch.uint8, device=d_idx.device)
        check(lib().dgrp_forward_windows(h, _ptr(d_idx), d_idx.numel(), step, w0, nw, _ptr(probs),
                                         _ptr(work), work.numel(), stream_ptr()), "dgrp_forward_windows")
        return probs

    def forward_windows_reference(self, d_idx: torch.Tensor, step: int, w0: int, nw: int) -> torch.Tensor:
        """The same windows through the plain-fp32 evaluation of the model on the device (dgrp_forward_windows_reference):
        the yardstick the fused fp16-operand kernel is measured against.  Slow; hundreds of windows."""
        probs = torch.empty((nw, self.vecsize, self.classes), dtype=torch.float32, device=d_idx.device)
        wb = lib().dgrp_forward_reference_workspace_bytes(self.handle, nw)
        work = torch.empty(max(wb, 256), dtype=torch.uint8, device=d_idx.device)
        check(lib().dgrp_forward_windows_reference(self.handle, _ptr(d_idx), d_idx.numel(), step, w0, nw, _ptr(probs),
                                                   _ptr(work), work.numel(), stream_ptr()), "dgrp_forward_windows_reference")
        return probs

    def check_accuracy(self, d_idx: Optional[torch.Tensor] = None, step: int = 50, windows: int = 256, seed: int = 0,
                       level: Optional[int] = None) -> dict:
        """Largest deviation of the fused kernel's class probabilities from the fp32 yardstick on up to `windows`
        windows spread evenly over the class-index tensor `d_idx` (default: a random ACGT sequence).  The bound the
        path is built to is 1e-3 (BASELINE north star).  `level` picks the fused kernel for the comparison (0 = fp16
        operands, 1 = split operands; default: the model's current setting) and is restored afterwards."""
        if level is not None:
            before = 1 if self.kernel_flags & 2 else 0
            self.set_precision(level)
            try:
                return self.check_accuracy(d_idx, step, windows, seed)
            finally:
                self.set_precision(before)
        dev = require_gpu()
        if d_idx is None:
            rng = np.random.default_rng(seed)
            d_idx = torch.from_numpy(rng.integers(0, 4, size=self.vecsize + step * windows, dtype=np.uint8)).to(dev)
        total = int(lib().dgrp_window_count(d_idx.numel(), self.vecsize, step))
        if total <= 0:
            raise ValueError(f"sequence of {d_idx.numel()} bases holds no window of {self.vecsize}")
        chunk = min(64, total)
        starts = sorted({int(x) for x in np.linspace(0, total - chunk, max(1, min(windows, total) // chunk))})
        worst, worst_window, checked, flips, above = 0.0, -1, 0, 0, 0
        per_window = []
        for w0 in starts:
            fast = self.forward_windows(d_idx, step, w0, chunk)
            ref = self.forward_windows_reference(d_idx, step, w0, chunk)
            dpos = (fast - ref).abs().amax(dim=2)                           # [chunk, T]
            diff = dpos.amax(dim=1)
            per_window.append(diff)
            k = int(diff.argmax())
            if float(diff[k]) > worst:
                worst, worst_window = float(diff[k]), w0 + k
            flips += int((fast.argmax(dim=2) != ref.argmax(dim=2)).sum())
            above += int((dpos > 1e-3).sum())
            checked += chunk
        pw = torch.cat(per_window).double()
        q = torch.quantile(pw, torch.tensor([0.5, 0.99], dtype=torch.float64, device=pw.device)).cpu().numpy()
        return {"max_abs_diff": worst, "window": worst_window, "windows_checked": checked, "argmax_flips": flips,
                "positions_checked": checked * self.vecsize, "positions_above_1e-3": above,
                "median_window_max": float(q[0]), "q99_window_max": float(q[1]),
                "windows_above_1e-3": int((pw > 1e-3).sum()), "within_1e-3": worst <= 1e-3}

    def predict_on_batch(self, batch) -> np.ndarray:
        """Keras-style call on a one-hot batch [b, T, 5]; returns numpy float32 [b, T, C]."""
        dev = require_gpu()
        x = torch.as_tensor(np.asarray(batch) if not isinstance(batch, torch.Tensor) else batch)
        if x.ndim != 3 or x.shape[1] != self.vecsize or x.shape[2] != 5:
            raise ValueError(f"expected a batch of shape [b, {self.vecsize}, 5], got {tuple(x.shape)}")
        x = x.to(dev)
        # the device path looks the input projection up by base (one-hot input): anything else -- soft labels, random
        # floats, an all-zero row -- would silently be read as the argmax base, where Keras computes the real projection
        if x.numel() and not bool((((x == 0) | (x == 1)).all(dim=2) & (x.sum(dim=2) == 1)).all()):
            raise ValueError("predict_on_batch takes one-hot batches (every [b, t, :] row holds a single 1): the device path looks "
                             "the input projection up by base")
        idx = x.argmax(dim=2).to(torch.uint8).reshape(-1).contiguous()
        b = x.shape[0]
        if b == 0:
            return np.zeros((0, self.vecsize, self.classes), np.float32)
        return self.forward_windows(idx, self.vecsize, 0, b).cpu().numpy()


def upload_sequence(raw: bytes) -> Tuple[int, torch.Tensor]:
    """A2 on the device: (startpos, class-index uint8 [N]).  Mirrors
    one_hot_encode_dna_sequence's stripping of leading/trailing 'N' (sequence.pyx:27-30),
    including the ValueError for an all-N record."""
    dev = require_gpu()
    st, kept = C.c_int64(0), C.c_int64(0)
    host = np.frombuffer(raw, dtype=np.uint8)
    check(lib().dgrp_strip_n(_np_ptr(host) if len(raw) else None, len(raw), C.byref(st), C.byref(kept)), "dgrp_strip_n")
    if kept.value < 0:
        raise ValueError("negative dimensions are not allowed")
    n = kept.value
    d_idx = torch.empty(n, dtype=torch.uint8, device=dev)
    if n:
        d_seq = torch.from_numpy(host[st.value:st.value + n].copy()).to(dev, non_blocking=False)
        check(lib().dgrp_encode(_ptr(d_seq), n, _ptr(d_idx), stream_ptr()), "dgrp_encode")
    return st.value, d_idx


class ContigPipeline:
    """Runs records through the device pipeline with the reference's CLI parameters."""

    def __init__(self, model: DeviceModel, step_size: int = 50, batch_size: int = 256, min_mss_len: int = 50,
                 xdrop_len: int = 50, use_mss: bool = True, chunk_windows: int = 1 << 20, precise: bool = False,
                 fast: bool = False, fp32: bool = False):
        self.model = model
        self.step = int(step_size)
        self.batch = int(batch_size)
        self.min_mss_len = int(min_mss_len)
        self.xdrop_len = int(xdrop_len)
        self.use_mss = bool(use_mss)
        self.chunk_windows = int(chunk_windows)
        # Which forward kernels run:
        #   default  split operands -- every model has such a fused kernel (class probabilities within 1e-5 of a float64
        #            evaluation in the tests, attention models included: their avg[t] crosses to the second kernel as float32);
        #   fast     fp16 operands (2-2.5x faster; 1e-3 on all but ill-conditioned windows);
        #   precise  accepted and equal to the default since round 2 made the default fp32-grade for every model (it used to send
        #            attention models through the 30 Mbp/s plain-fp32 kernels);
        #   fp32     the plain-fp32 kernels of ref_kernels.hip with the reference's own batch loop -- the yardstick, for tools.
        if precise and fast:
            raise ValueError("precise and fast exclude each other")
        self.precise, self.fast = bool(precise), bool(fast)
        self.fp32 = bool(fp32)
        self.split = not self.fast and not self.fp32
        self.event_log = None        # bench.py: list collecting (start, end, windows) per GRU launch
        self._view = None            # this pipeline's own handle on the model (dgrp_model_view): its precision level is never
                                     # changed, so pipelines of different levels can share a model on a pool of host threads
        if self.step < 1 or self.batch < 1:
            raise ValueError("step_size and batch_size must be >= 1")

    @property
    def handle(self):
        if self._view is None:
            self._view = self.model.view(1 if self.split else 0)
        return self._view

    def close(self) -> None:
        if self._view is not None:
            lib().dgrp_model_destroy(self._view)
            self._view = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # A3-A6
    def merged(self, d_idx: torch.Tensor) -> torch.Tensor:
        m, L = self.model, lib()
        n = d_idx.numel()
        out = torch.zeros((n, m.classes), dtype=torch.float32, device=d_idx.device)      # np.zeros, prediction.py:103
        nwin = L.dgrp_window_count(n, m.vecsize, self.step)
        if self.fp32:
            # the reference's own loop (prediction.py:104-110): batches of B windows, batch i lands at row i * b * step
            # (b = size of THAT batch); the forward pass runs for several batches at a time
            B = self.batch
            fit = (4 << 30) // (2 * m.vecsize * m.units * 4 + m.vecsize * (m.classes + 1) * 4)      # h_t of both strands, fp32
            per = max(B, min(self.chunk_windows, fit) // B * B)
            w0 = i = 0
            while w0 < nwin:
                nw = min(per, nwin - w0)
                probs = m.forward_windows_reference(d_idx, self.step, w0, nw)
                for off in range(0, nw, B):
                    b = min(B, nw - off)
                    index = i * b * self.step
                    if index < n:
                        check(L.dgrp_get_max(_ptr(out[index:]), n - index, _ptr(probs[off:off + b]), m.vecsize, m.classes,
                                             self.step, b, stream_ptr()), "dgrp_get_max")
                    i += 1
                w0 += nw
            return out
        if self.event_log is None and self.chunk_windows >= (1 << 20):
            # the library's own loop over the record (attention models with small spills: chunks alternating between its lanes)
            wb = L.dgrp_forward_merge_record_workspace_bytes(self.handle, n, self.step)
            work = torch.empty(max(wb, 256), dtype=torch.uint8, device=d_idx.device)
            check(L.dgrp_forward_merge_record(self.handle, _ptr(d_idx), n, self.step, self.batch, _ptr(out), _ptr(work), work.numel(),
                                              stream_ptr()), "dgrp_forward_merge_record")
            return out
        chunk = max(16, min(self.chunk_windows, L.dgrp_forward_window_chunk(self.handle)))     # attention: the avg[t] spill bounds a launch
        work = None
        w0 = 0
        while w0 < nwin:
            nw = min(chunk, nwin - w0)
            wb = L.dgrp_forward_workspace_bytes(self.handle, nw)
            if work is None or work.numel() < wb:
                work = torch.empty(max(wb, 256), dtype=torch.uint8, device=d_idx.device)
            if self.event_log is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            check(L.dgrp_forward_merge(self.handle, _ptr(d_idx), n, self.step, self.batch, w0, nw, _ptr(out),
                                       _ptr(work), work.numel(), stream_ptr()), "dgrp_forward_merge")
            if self.event_log is not None:
                ev1.record()
                self.event_log.append((ev0, ev1, nw))
            w0 += nw
        return out

    # A7-A10 (or A8)
    def labels(self, merged: torch.Tensor) -> torch.Tensor:
        L = lib()
        n, c = merged.shape
        dev = merged.device
        if n == 0:
            return torch.empty(0, dtype=torch.int8, device=dev)
        if self.use_mss:
            scores = torch.empty(n, dtype=torch.float64, device=dev)
            cls = torch.empty(n, dtype=torch.int8, device=dev)
            check(L.dgrp_scores(_ptr(merged), n, c, _ptr(scores), _ptr(cls), stream_ptr()), "dgrp_scores")
            return self.labels_from_scores(scores, cls)
        labels = torch.empty(n, dtype=torch.int8, device=dev)
        work = torch.empty(4096, dtype=torch.uint8, device=dev)
        check(L.dgrp_softmax_labels(_ptr(merged), n, c, None, _ptr(labels), _ptr(work), work.numel(),
                                    stream_ptr()), "dgrp_softmax_labels")
        return labels

    # A9-A10 on scores / classes that are already there (distributed.run_split assembles them from all ranks)
    def labels_from_scores(self, scores: torch.Tensor, cls: torch.Tensor) -> torch.Tensor:
        L = lib()
        n, dev = scores.numel(), scores.device
        labels = torch.empty(n, dtype=torch.int8, device=dev)
        if n == 0:
            return labels
        wb = L.dgrp_mss_workspace_bytes(n)
        work = torch.empty(wb, dtype=torch.uint8, device=dev)
        check(L.dgrp_mss_labels(_ptr(scores), _ptr(cls), n, self.model.classes, self.min_mss_len, self.xdrop_len, _ptr(labels),
                                None, _ptr(work), wb, stream_ptr()), "dgrp_mss_labels")
        return labels

    # A11
    def segments(self, labels: torch.Tensor, offset: int, contig: int = 0, cap: Optional[int] = None) -> np.ndarray:
        L = lib()
        n = labels.numel()
        dev = labels.device
        if n == 0:
            return np.zeros(0, SEGMENT_DTYPE)
        cap = cap if cap is not None else max(1024, n // 64)
        wb = L.dgrp_segments_workspace_bytes(n)
        work = torch.empty(wb, dtype=torch.uint8, device=dev)
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        while True:
            rec = torch.empty(cap * SEGMENT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            check(L.dgrp_segments(_ptr(labels), n, offset, contig, _ptr(rec), cap, _ptr(count), _ptr(work), wb,
                                  stream_ptr()), "dgrp_segments")
            total = int(count.item())
            if total <= cap:
                break
            cap = total                                     # rare: more segments than guessed, run again
        host = rec[: total * SEGMENT_DTYPE.itemsize].cpu().numpy()
        return host.view(SEGMENT_DTYPE).copy()

    def run_idx(self, d_idx: torch.Tensor, startpos: int, contig: int = 0) -> np.ndarray:
        """Segment records of one record whose class indices are on the device: one dgrp_predict_record call
        (the staged merged -> labels -> segments path is kept for callers that time or inspect the stages)."""
        if self.event_log is not None or self.fp32:
            return self.segments(self.labels(self.merged(d_idx)), startpos, contig)
        L = lib()
        n = d_idx.numel()
        if n == 0:
            return np.zeros(0, SEGMENT_DTYPE)
        dev = d_idx.device
        wb = L.dgrp_record_workspace_bytes(self.handle, n, self.step, int(self.use_mss))
        work = torch.empty(wb, dtype=torch.uint8, device=dev)
        cap = max(1024, n // 64)
        count = C.c_int64(0)
        while True:
            rec = torch.empty(cap * SEGMENT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            check(L.dgrp_predict_record(self.handle, _ptr(d_idx), n, self.step, self.batch, self.min_mss_len,
                                        self.xdrop_len, int(self.use_mss), int(startpos), int(contig), _ptr(rec), cap,
                                        C.byref(count), _ptr(work), wb, stream_ptr()), "dgrp_predict_record")
            if count.value <= cap:
                break
            cap = int(count.value)                          # rare: more segments than guessed, run again
        host = rec[: count.value * SEGMENT_DTYPE.itemsize].cpu().numpy()
        return host.view(SEGMENT_DTYPE).copy()

    def batchable(self) -> bool:
        """dgrp_predict_batch covers every model on the MSS path (the -m softmax is normalised per record)."""
        return self.use_mss and self.event_log is None and not self.fp32 and not getattr(self.model, "fp32_only", False)

    def run_batch(self, d_base: torch.Tensor, offsets, lengths, startposes, contigs) -> np.ndarray:
        """Segment records of MANY short records whose class indices lie in one device buffer (record r: `lengths[r]`
        >= 1 indices at `d_base[offsets[r]:]`): one dgrp_predict_batch call.  Rows come back in record order."""
        L = lib()
        nrec = len(lengths)
        if nrec == 0:
            return np.zeros(0, SEGMENT_DTYPE)
        off = np.ascontiguousarray(offsets, np.int64)
        ln = np.ascontiguousarray(lengths, np.int64)
        sp = np.ascontiguousarray(startposes, np.int64)
        cg = np.ascontiguousarray(contigs, np.int32)
        dev = d_base.device
        wb = L.dgrp_batch_workspace_bytes(self.handle, nrec, ln.ctypes.data, self.step)
        if wb <= 0:
            raise ValueError("run_batch: every record of a batch needs at least one base")
        work = torch.empty(wb, dtype=torch.uint8, device=dev)
        cap = max(1024, int(ln.sum()) // 64 + 2 * nrec)
        count = C.c_int64(0)
        while True:
            rec = torch.empty(cap * SEGMENT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            check(L.dgrp_predict_batch(self.handle, _ptr(d_base), nrec, off.ctypes.data, ln.ctypes.data, sp.ctypes.data,
                                       cg.ctypes.data, self.step, self.batch, self.min_mss_len, self.xdrop_len, _ptr(rec), cap,
                                       C.byref(count), _ptr(work), wb, stream_ptr()), "dgrp_predict_batch")
            if count.value <= cap:
                break
            cap = int(count.value)
        host = rec[: count.value * SEGMENT_DTYPE.itemsize].cpu().numpy()
        return host.view(SEGMENT_DTYPE).copy()

    def run(self, sequence, contig: int = 0) -> np.ndarray:
        raw = sequence.encode("utf-8") if isinstance(sequence, str) else bytes(sequence)
        startpos, d_idx = upload_sequence(raw)
        return self.run_idx(d_idx, startpos, contig)
