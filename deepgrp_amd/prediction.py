"""deepgrp_amd.prediction -- mirror of deepgrp/prediction.py:14-111 (the four functions on the
`deepgrp predict` path) over the HIP kernels.

`fetch_validation_batch` returns a WindowDataset instead of a tf.data.Dataset: iterating it
yields the same float32 [<=B, T, 5] batches; handing it to `predict` together with a model from
deepgrp_amd.model.load_model takes the fused device path (no window is ever materialised).
Any other (model, iterable) pair runs the reference's generic loop."""
from __future__ import annotations

from typing import Iterable, Iterator, Tuple

import numpy as np
import torch

from . import mss
from . import sequence as dgsequence
from ._lib import check, lib
from .model import Options
from .pipeline import ContigPipeline, DeviceModel, require_gpu, stream_ptr


class WindowDataset:
    """Sliding windows of a one-hot sequence (prediction.py:28-37): windows start at
    range(0, N - vecsize, step_size), batches of `batch_size`, the last one possibly shorter."""

    def __init__(self, data: np.ndarray, step_size: int, batch_size: int, vecsize: int):
        data = np.asarray(data)
        if data.ndim != 2:
            raise ValueError("data must be 2-dimensional [channels, N]")
        self.channels, self.n = int(data.shape[0]), int(data.shape[1])
        self.step_size, self.batch_size, self.vecsize = int(step_size), int(batch_size), int(vecsize)
        self._host = data
        self.nwin = len(range(0, self.n - self.vecsize, self.step_size))
        self._d_idx = None

    @property
    def element_shape(self) -> Tuple[None, int, int]:
        return (None, self.vecsize, self.channels)

    def device_index(self) -> torch.Tensor:
        """Class index per base on the GPU (argmax over the one-hot channels)."""
        if self._d_idx is None:
            if self.channels != 5:
                raise ValueError("the device path needs the 5-channel one-hot encoding")
            dev = require_gpu()
            self._d_idx = torch.from_numpy(np.ascontiguousarray(self._host)).to(dev).argmax(dim=0).to(torch.uint8).contiguous()
        return self._d_idx

    def __len__(self) -> int:
        return (self.nwin + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[np.ndarray]:
        if self.channels == 5 and set(np.unique(self._host)) <= {0, 1}:
            d_idx = self.device_index()
            for w0 in range(0, self.nwin, self.batch_size):
                nw = min(self.batch_size, self.nwin - w0)
                out = torch.empty((nw, self.vecsize, 5), dtype=torch.float32, device=d_idx.device)
                check(lib().dgrp_windows_onehot(d_idx.data_ptr(), self.n, self.vecsize, self.step_size, w0, nw, 4,
                                                out.data_ptr(), stream_ptr()), "dgrp_windows_onehot")
                yield out.cpu().numpy()
        else:                                   # arbitrary matrices (the reference's own test feeds random floats)
            dt = self._host.T
            starts = list(range(0, self.n - self.vecsize, self.step_size))
            for b0 in range(0, len(starts), self.batch_size):
                yield np.stack([dt[i:i + self.vecsize].astype("float32") for i in starts[b0:b0 + self.batch_size]])

    as_numpy_iterator = __iter__


def fetch_validation_batch(data: np.ndarray, step_size: int, batch_size: int, vecsize: int) -> WindowDataset:
    """Function to fetch a validation batch (no randomization), prediction.py:14-37."""
    return WindowDataset(data, step_size, batch_size, vecsize)


def predict(model, data: Iterable, results_shape: Tuple[int, int], step_size: int) -> np.ndarray:
    """Predict for complete data using an iterator over all data (prediction.py:89-111): zeros
    [N, C], every batch max-merged at `i * batch.shape[0] * step_size` -- including the offset the
    reference gets for a short last batch."""
    if isinstance(model, DeviceModel) and isinstance(data, WindowDataset) and data.channels == 5 \
            and data.vecsize == model.vecsize and tuple(results_shape) == (data.n, model.classes) \
            and data.step_size == step_size:
        pipe = ContigPipeline(model, step_size, data.batch_size)
        return pipe.merged(data.device_index()).cpu().numpy()
    predictions = np.zeros(results_shape, dtype=np.float32)
    for i, batch in enumerate(data):
        index = i * batch.shape[0] * step_size
        probas = model.predict_on_batch(batch)
        try:
            dgsequence.get_max(predictions[index:], probas, step_size)
        except TypeError:
            dgsequence.get_max(predictions[index:], probas.numpy(), step_size)
    return predictions


def _scores_and_classes(probs: np.ndarray):
    dev = require_gpu()
    d_p = torch.from_numpy(np.ascontiguousarray(probs, dtype=np.float32)).to(dev)
    n, c = d_p.shape
    d_s = torch.empty(n, dtype=torch.float64, device=dev)
    d_c = torch.empty(n, dtype=torch.int8, device=dev)
    check(lib().dgrp_scores(d_p.data_ptr(), n, c, d_s.data_ptr(), d_c.data_ptr(), stream_ptr()), "dgrp_scores")
    return d_s.cpu().numpy(), d_c.cpu().numpy().astype(np.int64)


def apply_mss(probs: np.ndarray, options: Options) -> np.ndarray:
    """Applies the maximum scoring segment algorithm to probabilities [N, C] (prediction.py:40-59);
    returns the one-hot float64 [N, C] array of the reference."""
    probs = np.asarray(probs)
    nof_labels = probs.shape[1]
    scores, results_classes = _scores_and_classes(probs)
    return mss.find_mss_labels(scores, results_classes, nof_labels, options.min_mss_len, options.xdrop_len)


def softmax(array: np.ndarray) -> np.ndarray:
    """Softmax with the global maximum subtracted (prediction.py:62-65)."""
    array = np.asarray(array)
    dev = require_gpu()
    if array.dtype != np.float32 or array.ndim != 2 or array.shape[1] > 16:
        # other dtypes / wider rows: same expression in the input precision on the device
        x = torch.from_numpy(np.ascontiguousarray(array)).to(dev)
        e_x = torch.exp(x - x.max())
        return (e_x / e_x.sum(dim=1, keepdim=True)).cpu().numpy()
    d_p = torch.from_numpy(np.ascontiguousarray(array)).to(dev)
    n, c = d_p.shape
    d_sm = torch.empty((n, c), dtype=torch.float32, device=dev)
    d_l = torch.empty(n, dtype=torch.int8, device=dev)
    work = torch.empty(4096, dtype=torch.uint8, device=dev)
    check(lib().dgrp_softmax_labels(d_p.data_ptr(), n, c, d_sm.data_ptr(), d_l.data_ptr(), work.data_ptr(), 4096,
                                    stream_ptr()), "dgrp_softmax_labels")
    return d_sm.cpu().numpy()
