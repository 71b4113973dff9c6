"""deepgrp_amd.preprocessing -- mirror of deepgrp/preprocessing.py (the label-side data
formats next to the prediction path): annotations written by parse_rm.py -> one-hot truth, trimming of the
leading/trailing N block, and the `Data` pair predict_complete takes."""
from __future__ import annotations

import os
from typing import List, NamedTuple, Tuple

import numpy as np


def preprocess_y(filename: os.PathLike, chromosom: str, length: int, repeats_to_search: List[int]) -> np.ndarray:
    """One-hot int8 [len(repeats_to_search) + 1, length] truth from the whitespace separated table of
    parse_rm.py (columns: contig, begin, end, repeat number, ...), deepgrp/preprocessing.py:9-48.

    Like the reference, a kept row sets `y[repeatnumber, begin:end] = 1` -- the row index is the repeat NUMBER,
    not its position in `repeats_to_search`, so a number >= the row count raises IndexError -- and row 0 marks
    the bases no kept repeat covers."""
    begins, ends, numbers = [], [], []
    wanted = set(int(r) for r in repeats_to_search)
    with open(filename, "r") as fh:
        for line in fh:
            cols = line.split()
            if not cols:
                continue
            if cols[0] != chromosom:
                continue
            number = int(cols[3])
            if number in wanted:
                begins.append(int(cols[1]))
                ends.append(int(cols[2]))
                numbers.append(number)
    yarray = np.zeros((len(repeats_to_search) + 1, length), dtype=np.int8)
    for b, e, r in zip(begins, ends, numbers):
        yarray[r, b:e] = 1
    yarray[0, yarray[1:].sum(axis=0) == 0] = 1
    return yarray


def drop_start_end_n(fwd: np.ndarray, array: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Slices of both arguments without the leading and trailing N block (deepgrp/preprocessing.py:51-70).
    The reference's end index is that of the LAST non-N base, used as an exclusive bound: that base is dropped
    as well -- kept."""
    sums = fwd[0:4].sum(axis=0)
    start = np.argmax(sums > 0)
    end = fwd.shape[1] - 1 - np.argmax(np.flip(sums) > 0)
    return fwd[:, start:end], array[:, start:end]


# Collection of forward one hot encoded sequence and true annotations (deepgrp/preprocessing.py:73-74)
Data = NamedTuple("Data", [("fwd", np.ndarray), ("truelbl", np.ndarray)])


def load_onehot_npz(path: os.PathLike) -> np.ndarray:
    """The `fwd` array (int8 [5, N]) of a `<fasta>.gz.npz` file written by preprocess_sequence.py
    (deepgrp/_scripts/preprocess_sequence.py:71-78; read back in deepgrp/__main__.py's training command)."""
    with np.load(path, allow_pickle=False) as z:
        fwd = z["fwd"]
    if fwd.ndim != 2 or fwd.shape[0] != 5:
        raise ValueError(f"{path}: expected a one-hot array of shape [5, N], found {fwd.shape}")
    return fwd
