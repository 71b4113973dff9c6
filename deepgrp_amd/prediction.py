"""deepgrp_amd.prediction -- mirror of deepgrp/prediction.py over the HIP kernels: the four functions on the
`deepgrp predict` path (:14-111) and the evaluation surface next to it (predict_complete :114-141, metrics
:144-241, filter_segments :244-260).

`fetch_validation_batch` returns a WindowDataset instead of a tf.data.Dataset: iterating it
yields the same float32 [<=B, T, 5] batches; handing it to `predict` together with a model from
deepgrp_amd.model.load_model takes the fused device path (no window is ever materialised).
Any other (model, iterable) pair runs the reference's generic loop."""
from __future__ import annotations

import os
from typing import Dict, Iterable, Iterator, Tuple, Union

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
            d = torch.from_numpy(np.ascontiguousarray(self._host)).to(dev)
            # one-hot columns only (what one_hot_encode_dna_sequence produces): the device path looks the input projection up by base
            if d.numel() and not bool((((d == 0) | (d == 1)).all(dim=0) & (d.sum(dim=0) == 1)).all()):
                raise ValueError("the device path takes one-hot data (every column holds a single 1)")
            self._d_idx = d.argmax(dim=0).to(torch.uint8).contiguous()
        return self._d_idx

    def __len__(self) -> int:
        return (self.nwin + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[np.ndarray]:
        if self.channels == 5 and set(np.unique(self._host)) <= {0, 1} and bool((self._host.sum(axis=0) == 1).all()):
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


# ------------------------------------------------------------------------------------------------
# N2: predict_complete and the evaluation helpers (prediction.py:68-86, :114-260)
# ------------------------------------------------------------------------------------------------
def setup_prediction_from_options_checkpoint(options: Options, logdir) -> DeviceModel:
    """prediction.py:68-86 restores the latest TensorFlow checkpoint under `logdir` into a freshly created
    model.  TensorFlow's checkpoint bundles are not readable here; the weights are taken from a Keras HDF5 file
    instead: `logdir` is either that file or a directory holding one (`*.h5` / `*.hdf5`, the newest wins).
    The file's layer graph must agree with `options` (vecsize, units, attention, rnn)."""
    from .model import ModelFormatError, load_model
    path = os.fspath(logdir)
    if os.path.isdir(path):
        cands = [os.path.join(path, f) for f in os.listdir(path) if f.endswith((".h5", ".hdf5"))]
        if not cands:
            raise ModelFormatError(f"{path}: no Keras HDF5 model (*.h5, *.hdf5) found; TensorFlow checkpoint "
                                   "bundles cannot be read without TensorFlow -- export the model with model.save()")
        path = max(cands, key=os.path.getmtime)
    model = load_model(path, custom_objects={"ReverseComplement": None})
    for name, have in (("vecsize", model.vecsize), ("units", model.units), ("attention", bool(model.attention))):
        want = options[name]
        if (bool(want) if name == "attention" else int(want)) != have:
            raise ModelFormatError(f"{path}: {name}={have} in the file but {want} in the options")
    return model


def predict_complete(step_size: int, options: Options, logdir, data, use_mss: bool = False) -> np.ndarray:
    """Restores a model and predicts for a sequence (prediction.py:114-141): `data` has `.fwd` (one-hot
    [5, N]) and `.truelbl` ([C, N]); returns apply_mss's one-hot labels or the softmax of the merged
    probabilities."""
    model = setup_prediction_from_options_checkpoint(options, logdir)
    val_iterator = fetch_validation_batch(data.fwd, step_size, options.batch_size, options.vecsize)
    output_shape = data.truelbl.shape[::-1]
    predictions = predict(model, val_iterator, output_shape, step_size)
    if use_mss:
        return apply_mss(predictions, options)
    return softmax(predictions)


def calculate_multiclass_matthews_cc(cnf_matrix: np.ndarray) -> float:
    """R_K / multi-class Matthews correlation coefficient of a confusion matrix (prediction.py:144-162)."""
    cnf_matrix = np.asarray(cnf_matrix)
    t_sum = cnf_matrix.sum(axis=1, dtype=float)
    p_sum = cnf_matrix.sum(axis=0, dtype=float)
    n_correct = np.trace(cnf_matrix, dtype=float)
    n_samples = p_sum.sum()
    cov_ytyp = n_correct * n_samples - np.dot(t_sum, p_sum)
    cov_ypyp = n_samples**2 - np.dot(p_sum, p_sum)
    cov_ytyt = n_samples**2 - np.dot(t_sum, t_sum)
    return cov_ytyp / np.sqrt(cov_ytyt * cov_ypyp)


def _calculate_metrics(cnf_matrix: np.ndarray) -> Dict[str, Union[np.ndarray, float]]:
    """Per-class rates from a confusion matrix, same keys and formulas as prediction.py:165-201."""
    cnf_matrix = np.asarray(cnf_matrix)
    tp = np.diag(cnf_matrix).astype(float)
    fp = (cnf_matrix.sum(axis=0) - tp).astype(float)
    fn = (cnf_matrix.sum(axis=1) - tp).astype(float)
    tn = (cnf_matrix.sum() - (fp + fn + tp)).astype(float)
    metrics: Dict[str, Union[np.ndarray, float]] = {}
    metrics["TPR"] = tp / (tp + fn)
    metrics["TNR"] = tn / (tn + fp)
    metrics["PPV"] = tp / (tp + fp)
    metrics["NPV"] = tn / (tn + fn)
    metrics["FPR"] = fp / (fp + tn)
    metrics["FNR"] = fn / (tp + fn)
    metrics["FDR"] = fp / (tp + fp)
    metrics["ACC"] = (tp + tn) / (tp + fp + fn + tn)
    metrics["F1"] = 2 * metrics["TPR"] * metrics["PPV"] / (metrics["TPR"] + metrics["PPV"])
    metrics["MCC"] = calculate_multiclass_matthews_cc(cnf_matrix)
    return metrics


def _device_labels(a) -> "torch.Tensor | None":
    """int8 device copy of a label array when it can go through the kernels (values within int8), else None."""
    if isinstance(a, torch.Tensor):
        if a.is_cuda and a.dtype == torch.int8 and a.dim() == 1:
            return a.contiguous()
        a = a.cpu().numpy()
    a = np.asarray(a)
    if a.ndim != 1 or a.size == 0 or a.dtype.kind not in "iu":
        return None
    if a.min() < -128 or a.max() > 127:
        return None
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.int8))).to(require_gpu())


def confusion_matrix(truelbl, predictedlbl) -> np.ndarray:
    """Confusion matrix from integer label arrays (prediction.py:204-222); numpy arrays or int8 CUDA tensors.
    n_classes = (max over both) - (min over both) + 1 and the cells are indexed with the raw labels, exactly as
    the reference does; label sets it cannot index raise IndexError there and here."""
    t_size = truelbl.numel() if isinstance(truelbl, torch.Tensor) else np.asarray(truelbl).size
    p_size = predictedlbl.numel() if isinstance(predictedlbl, torch.Tensor) else np.asarray(predictedlbl).size
    assert t_size == p_size
    d_t, d_p = _device_labels(truelbl), _device_labels(predictedlbl)
    if d_t is None or d_p is None:
        raise TypeError("confusion_matrix needs one-dimensional, non-empty integer label arrays within int8 range")
    lo = min(int(d_t.min()), int(d_p.min()))
    hi = max(int(d_t.max()), int(d_p.max()))
    k = hi - lo + 1
    if lo < -k or hi >= k:
        raise IndexError(f"index {hi if hi >= k else lo} is out of bounds for axis 0 with size {k}")
    if k > 16:
        raise ValueError("confusion_matrix: more than 16 classes are not supported on the device")
    if lo < 0:                                              # numpy wraps negative indices
        d_t = torch.where(d_t < 0, d_t + k, d_t)
        d_p = torch.where(d_p < 0, d_p + k, d_p)
    d_cnf = torch.empty((k, k), dtype=torch.int64, device=d_t.device)
    d_bad = torch.empty(1, dtype=torch.int32, device=d_t.device)
    check(lib().dgrp_confusion_matrix(d_t.data_ptr(), d_p.data_ptr(), d_t.numel(), k, d_cnf.data_ptr(), d_bad.data_ptr(),
                                      stream_ptr()), "dgrp_confusion_matrix")
    assert int(d_bad.item()) == 0
    return d_cnf.cpu().numpy().astype(int)


def calculate_metrics(predictions_class, true_class) -> Tuple[np.ndarray, Dict[str, Union[np.ndarray, float]]]:
    """Confusion matrix and the metrics dictionary incl. TotalACC (prediction.py:225-241)."""
    cnf_matrix = confusion_matrix(true_class, predictions_class)
    n = true_class.numel() if isinstance(true_class, torch.Tensor) else np.asarray(true_class).shape[0]
    if isinstance(true_class, torch.Tensor) or isinstance(predictions_class, torch.Tensor):
        a, b = _device_labels(true_class), _device_labels(predictions_class)
        overall_acc = int((a == b).sum()) / n
    else:
        overall_acc = (np.asarray(true_class) == np.asarray(predictions_class)).sum() / n
    metrics = _calculate_metrics(cnf_matrix)
    metrics["TotalACC"] = overall_acc
    return cnf_matrix, metrics


def filter_segments(array, min_len: int = 50) -> None:
    """Clear runs of one positive label shorter than `min_len`, in place (prediction.py:244-260).  Accepts a
    numpy array (any dtype whose values fit int8; written back in place) or an int8 CUDA tensor."""
    if isinstance(array, torch.Tensor) and array.is_cuda:
        if array.dtype != torch.int8 or array.dim() != 1 or not array.is_contiguous():
            raise TypeError("device labels must be a contiguous one-dimensional int8 tensor")
        check(lib().dgrp_filter_segments(array.data_ptr(), array.data_ptr(), array.numel(), int(min_len), stream_ptr()),
              "dgrp_filter_segments")
        return
    a = np.asarray(array)
    if a.ndim != 1:
        raise ValueError("filter_segments works on one-dimensional label arrays")
    if a.size == 0:
        return
    vals = a if a.dtype.kind in "iu" else None
    if vals is None:
        if not np.all(a == np.round(a)):
            raise TypeError("filter_segments needs integer-valued labels")
    if a.min() < -128 or a.max() > 127:
        raise ValueError("labels outside the int8 range are not supported on the device")
    dev = require_gpu()
    d = torch.from_numpy(np.ascontiguousarray(a.astype(np.int8))).to(dev)
    out = torch.empty_like(d)
    check(lib().dgrp_filter_segments(d.data_ptr(), out.data_ptr(), d.numel(), int(min_len), stream_ptr()),
          "dgrp_filter_segments")
    keep = out.cpu().numpy() != 0
    array[~keep & (a > 0)] = 0
