"""Error behaviour of the C ABI (include/deepgrp_hip.h): bad arguments come back as DGRP_EINVAL with a message in
dgrp_last_error(), before any device work -- so this half runs without a GPU.  The reference raises
ValueError/TypeError from its Cython argument checks at the same places (deepgrp/sequence.pyx:25-36,
deepgrp/mss.pyx:24-27); the mirror package turns these codes into exceptions."""
import ctypes as C

import pytest

from deepgrp_amd._lib import lib

EINVAL = -1
P = 0x10000                    # a non-NULL pointer value that is never dereferenced: an earlier check fails
i64x1 = (C.c_int64 * 1)(0)


def info4():
    return (C.c_int64 * 4)()


CASES = [
    ("dgrp_strip_n", lambda: (None, 5, i64x1, i64x1), "dgrp_strip_n"),
    ("dgrp_strip_n", lambda: (P, -1, i64x1, i64x1), "dgrp_strip_n"),
    ("dgrp_encode", lambda: (None, 10, P, None), "dgrp_encode"),
    ("dgrp_encode", lambda: (P, -1, P, None), "dgrp_encode"),
    ("dgrp_onehot", lambda: (P, 3, None, None), "dgrp_onehot"),
    ("dgrp_fasta_encode", lambda: (P, -1, P, info4(), P, 0, None), "dgrp_fasta_encode"),
    ("dgrp_fasta_encode", lambda: (P, 10, P, None, P, 1 << 20, None), "dgrp_fasta_encode"),
    ("dgrp_fasta_encode", lambda: (None, 10, P, info4(), P, 1 << 20, None), "NULL pointer"),
    ("dgrp_fasta_encode_batch", lambda: (P, -1, None, None, P, None, P, 0, None), "dgrp_fasta_encode_batch"),
    ("dgrp_fasta_encode_batch", lambda: (P, 2, None, None, P, None, P, 0, None), "dgrp_fasta_encode_batch"),
    ("dgrp_windows_onehot", lambda: (P, 100, 10, 5, 0, 1, 3, P, None), "elem"),
    ("dgrp_windows_onehot", lambda: (P, 100, 0, 5, 0, 1, 4, P, None), "bad T/s/w0/nw"),
    ("dgrp_windows_onehot", lambda: (P, 100, 10, 0, 0, 1, 4, P, None), "bad T/s/w0/nw"),
    ("dgrp_windows_onehot", lambda: (None, 100, 10, 5, 0, 1, 4, P, None), "NULL pointer"),
    ("dgrp_windows_onehot", lambda: (P, 100, 10, 5, 18, 2, 4, P, None), "runs past"),
    ("dgrp_model_create", lambda: (None, 200, 32, 5, 0, P, P, P, P, P, None), "NULL out"),
    ("dgrp_model_create", lambda: (C.pointer(C.c_void_p()), 0, 32, 5, 0, P, P, P, P, P, None), "window size"),
    ("dgrp_model_create", lambda: (C.pointer(C.c_void_p()), 70000, 32, 5, 0, P, P, P, P, P, None), "window size"),
    ("dgrp_model_create", lambda: (C.pointer(C.c_void_p()), 200, 0, 5, 0, P, P, P, P, P, None), "units"),
    ("dgrp_model_create", lambda: (C.pointer(C.c_void_p()), 200, 2049, 5, 0, P, P, P, P, P, None), "units"),
    ("dgrp_model_create", lambda: (C.pointer(C.c_void_p()), 200, 32, 1, 0, P, P, P, P, P, None), "classes"),
    ("dgrp_model_create", lambda: (C.pointer(C.c_void_p()), 200, 32, 65, 0, P, P, P, P, P, None), "classes"),
    ("dgrp_model_create", lambda: (C.pointer(C.c_void_p()), 200, 32, 5, 0, P, None, P, None, P, P), "NULL tensor"),
    ("dgrp_model_create", lambda: (C.pointer(C.c_void_p()), 200, 32, 5, 1, P, P, P, None, P, P), "NULL tensor"),   # attention without a scale
    ("dgrp_model_create_lstm", lambda: (None, 200, 32, 5, P, P, P, P, P), "NULL out"),
    ("dgrp_model_create_lstm", lambda: (C.pointer(C.c_void_p()), 200, 2049, 5, P, P, P, P, P), "units"),
    ("dgrp_model_create_lstm", lambda: (C.pointer(C.c_void_p()), 200, 32, 5, P, P, None, P, P), "NULL tensor"),
    ("dgrp_model_dims", lambda: (None, None, None, None, None), "NULL model"),
    ("dgrp_model_flags", lambda: (None,), "NULL model"),
    ("dgrp_forward_windows", lambda: (None, P, 1000, 50, 0, 1, P, P, 0, None), "dgrp_forward"),
    ("dgrp_forward_merge", lambda: (None, P, 1000, 50, 256, 0, 1, P, P, 0, None), "dgrp_forward"),
    ("dgrp_forward_windows_reference", lambda: (None, P, 1000, 50, 0, 1, P, P, 1 << 20, None), "bad arguments"),
    ("dgrp_get_max", lambda: (P, 10, P, 0, 5, 5, 1, None), "bad shape"),
    ("dgrp_get_max", lambda: (P, 10, P, 2, 5, 0, 1, None), "bad shape"),
    ("dgrp_get_max", lambda: (None, 10, P, 2, 5, 5, 1, None), "NULL pointer"),
    ("dgrp_scores", lambda: (P, -1, 5, P, P, None), "bad n/C"),
    ("dgrp_scores", lambda: (P, 10, 0, P, P, None), "bad n/C"),
    ("dgrp_scores", lambda: (P, 10, 65, P, P, None), "bad n/C"),
    ("dgrp_scores", lambda: (P, 10, 5, None, P, None), "NULL pointer"),
    ("dgrp_softmax_labels", lambda: (P, 10, 0, P, P, P, 1 << 20, None), "bad n/C"),
    ("dgrp_softmax_labels", lambda: (P, 10, 5, P, P, P, 16, None), "workspace"),
    ("dgrp_mss_labels", lambda: (P, P, -1, 5, 50, 50, P, None, P, 1 << 30, None), "out of range"),
    ("dgrp_mss_labels", lambda: (P, P, 1 << 31, 5, 50, 50, P, None, P, 1 << 30, None), "out of range"),
    ("dgrp_mss_labels", lambda: (P, P, 100, 1, 50, 50, P, None, P, 1 << 30, None), "nof_labels"),
    ("dgrp_mss_labels", lambda: (P, P, 100, 65, 50, 50, P, None, P, 1 << 30, None), "nof_labels"),
    ("dgrp_mss_labels", lambda: (None, P, 100, 5, 50, 50, P, None, P, 1 << 30, None), "NULL pointer"),
    ("dgrp_mss_segments_host", lambda: (None, 0, None, 0, None), "NULL pointer"),
    ("dgrp_mss_labels_batch", lambda: (P, P, -1, 1, i64x1, 5, 50, 50, P, P, 1 << 30, None), "bad arguments"),
    ("dgrp_mss_labels_batch", lambda: (P, P, 128, 2, None, 5, 50, 50, P, P, 1 << 30, None), "bad arguments"),
    ("dgrp_mss_labels_batch", lambda: (P, P, 128, 2, (C.c_int64 * 3)(0, 64, 128), 1, 50, 50, P, P, 1 << 30, None), "nof_labels"),
    ("dgrp_mss_labels_batch", lambda: (P, P, 128, 2, (C.c_int64 * 3)(0, 64, 100), 5, 50, 50, P, P, 1 << 30, None), "from 0 to total_n"),
    ("dgrp_mss_labels_batch", lambda: (P, P, 128, 2, (C.c_int64 * 3)(0, 50, 128), 5, 50, 50, P, P, 1 << 30, None), "multiples of 64"),
    ("dgrp_mss_labels_batch", lambda: (P, P, 128, 2, (C.c_int64 * 3)(0, 0, 128), 5, 50, 50, P, P, 1 << 30, None), "increasing"),
    ("dgrp_segments", lambda: (P, -1, 0, 0, P, 10, P, P, 1 << 20, None), "bad arguments"),
    ("dgrp_segments", lambda: (P, 10, 0, 0, P, 10, None, P, 1 << 20, None), "bad arguments"),
    ("dgrp_segments", lambda: (None, 10, 0, 0, P, 10, P, P, 1 << 20, None), "NULL pointer"),
    ("dgrp_predict_record", lambda: (None, P, 1000, 50, 256, 50, 50, 1, 0, 0, P, 10, i64x1, P, 1 << 30, None), "bad arguments"),
    ("dgrp_predict_batch", lambda: (None, P, 1, P, P, P, P, 50, 256, 50, 50, P, 10, i64x1, P, 1 << 30, None), "bad arguments"),
    ("dgrp_confusion_matrix", lambda: (P, P, 10, 0, P, P, None), "classes"),
    ("dgrp_confusion_matrix", lambda: (P, P, 10, 65, P, P, None), "classes"),
    ("dgrp_confusion_matrix", lambda: (P, P, -1, 5, P, P, None), "bad arguments"),
    ("dgrp_confusion_matrix", lambda: (P, P, 10, 5, None, P, None), "bad arguments"),
    ("dgrp_filter_segments", lambda: (P, P, -1, 50, None), "negative length"),
    ("dgrp_filter_segments", lambda: (None, P, 10, 50, None), "NULL pointer"),
    ("dgrp_filter_segments", lambda: (P, P, 1 << 39, 50, None), "too long"),
]


@pytest.mark.parametrize("name,make,fragment", CASES, ids=[f"{c[0]}-{i}" for i, c in enumerate(CASES)])
def test_bad_arguments_are_refused_before_any_device_work(name, make, fragment):
    L = lib()
    rc = getattr(L, name)(*make())
    assert rc == EINVAL, (name, rc)
    msg = L.dgrp_last_error().decode()
    assert fragment in msg, msg


ENOMEM = -3
SMALL_WORKSPACE = [
    ("dgrp_fasta_encode", lambda: (P, 1000, P, info4(), P, 8, None)),
    ("dgrp_fasta_encode_batch", lambda: (P, 2, (C.c_int64 * 2)(0, 500), (C.c_int64 * 2)(500, 500), P, (C.c_int64 * 8)(), P, 8, None)),
    ("dgrp_mss_labels", lambda: (P, P, 1000, 5, 50, 50, P, None, P, 8, None)),
    ("dgrp_mss_labels_batch", lambda: (P, P, 128, 2, (C.c_int64 * 3)(0, 64, 128), 5, 50, 50, P, P, 8, None)),
    ("dgrp_segments", lambda: (P, 1000, 0, 0, P, 10, P, P, 8, None)),
]


@pytest.mark.parametrize("name,make", SMALL_WORKSPACE, ids=[c[0] for c in SMALL_WORKSPACE])
def test_short_workspace_is_refused_before_any_device_work(name, make):
    """The sizes come from the matching *_workspace_bytes call; a shorter buffer is DGRP_ENOMEM, never a partial run."""
    L = lib()
    assert getattr(L, name)(*make()) == ENOMEM
    assert "workspace" in L.dgrp_last_error().decode()


def test_size_queries_are_total_functions():
    """The *_bytes / count helpers take any int64 without failing; non-positive sizes give a small non-negative answer."""
    L = lib()
    for n in (0, 1, 199, 200, 201, 249, 250, 251, 1000, 10 ** 6 + 7):      # deepgrp/prediction.py:31: range(0, n - T, s)
        assert L.dgrp_window_count(n, 200, 50) == len(range(0, n - 200, 50)), n
    assert L.dgrp_window_count(1000, 0, 50) == 0 and L.dgrp_window_count(1000, 200, 0) == 0
    for f, args in (("dgrp_fasta_workspace_bytes", (0,)), ("dgrp_fasta_batch_workspace_bytes", (0, 0)),
                    ("dgrp_mss_workspace_bytes", (0,)), ("dgrp_mss_batch_workspace_bytes", (0, 0)),
                    ("dgrp_segments_workspace_bytes", (0,))):
        assert getattr(L, f)(*args) >= 0, f
    # monotone in n: a caller may size for the largest record once
    for f in ("dgrp_fasta_workspace_bytes", "dgrp_mss_workspace_bytes", "dgrp_segments_workspace_bytes"):
        sizes = [getattr(L, f)(n) for n in (1, 1000, 10 ** 6, 10 ** 8, 2 ** 31 - 1)]
        assert sizes == sorted(sizes) and sizes[0] > 0, (f, sizes)
