"""deepgrp_amd.mss -- mirror of the reference's Cython module deepgrp/_mss/pymss.pyx
(stub deepgrp/mss.pyi) over the HIP maximal-scoring-segment kernels."""
from __future__ import annotations

import numpy as np
import torch

from ._lib import check, lib
from .pipeline import require_gpu, stream_ptr


def find_mss_labels(inputs: np.ndarray, label: np.ndarray, nof_labels: int, min_mss_len: int,
                    xdrop_len: int) -> np.ndarray:
    """Maximum scoring segments with labels (pymss.pyx:16-27): float64 scores [n], integer labels
    [n] -> float64 one-hot [n, nof_labels].  `None` arguments raise TypeError (`not None` in the
    Cython signature); wrong dtypes raise ValueError."""
    if inputs is None or label is None:
        raise TypeError("Argument 'inputs'/'label' must not be None")
    inputs = np.asarray(inputs)
    label = np.asarray(label)
    if inputs.dtype != np.float64:
        raise ValueError(f"Buffer dtype mismatch, expected 'double' but got '{inputs.dtype}'")
    if label.dtype != np.int64:
        raise ValueError(f"Buffer dtype mismatch, expected 'long' but got '{label.dtype}'")
    if inputs.ndim != 1 or label.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1)")
    n = inputs.shape[0]
    out = np.zeros((n, nof_labels))
    if n == 0:
        return out
    dev = require_gpu()
    d_s = torch.from_numpy(np.ascontiguousarray(inputs)).to(dev)
    d_l = torch.from_numpy(label.astype(np.int8)).to(dev)
    d_o = torch.empty(n, dtype=torch.int8, device=dev)
    wb = lib().dgrp_mss_workspace_bytes(n)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    check(lib().dgrp_mss_labels(d_s.data_ptr(), d_l.data_ptr(), n, int(nof_labels), int(min_mss_len), int(xdrop_len),
                                d_o.data_ptr(), None, work.data_ptr(), wb, stream_ptr()), "dgrp_mss_labels")
    out[np.arange(n), d_o.cpu().numpy().astype(np.int64)] = 1.0
    return out
