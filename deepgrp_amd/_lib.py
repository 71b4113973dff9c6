"""ctypes binding of libdeepgrp_hip.so (the C ABI declared in include/deepgrp_hip.h)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdeepgrp_hip.so")

i64, i32, vp, cint = C.c_int64, C.c_int32, C.c_void_p, C.c_int


class DgrpError(RuntimeError):
    """A libdeepgrp_hip call failed (message from dgrp_last_error())."""


class Segment(C.Structure):
    """struct dgrp_segment"""
    _fields_ = [("start", C.c_int64), ("end", C.c_int64), ("label", C.c_int32), ("contig", C.c_int32)]


_SIGNATURES = {
    "dgrp_abi_version": (cint, []),
    "dgrp_last_error": (C.c_char_p, []),
    "dgrp_device_info": (cint, [C.c_char_p, C.c_size_t, C.POINTER(cint), C.POINTER(i64)]),
    "dgrp_strip_n": (cint, [vp, i64, C.POINTER(i64), C.POINTER(i64)]),
    "dgrp_encode": (cint, [vp, i64, vp, vp]),
    "dgrp_onehot": (cint, [vp, i64, vp, vp]),
    "dgrp_fasta_workspace_bytes": (i64, [i64]),
    "dgrp_fasta_encode": (cint, [vp, i64, vp, C.POINTER(i64), vp, i64, vp]),
    "dgrp_fasta_batch_workspace_bytes": (i64, [i64, i64]),
    "dgrp_fasta_encode_batch": (cint, [vp, i64, vp, vp, vp, vp, vp, i64, vp]),
    "dgrp_fasta_chunks_workspace_bytes": (i64, [i64]),
    "dgrp_fasta_chunks": (cint, [vp, i64, i64, vp, vp, C.POINTER(i64), vp, i64, vp]),
    "dgrp_format_rows_bound": (i64, [i64, i64]),
    "dgrp_format_rows": (cint, [vp, vp, i64, cint, vp, i64, vp, i64, C.POINTER(i64)]),
    "dgrp_window_count": (i64, [i64, i64, i64]),
    "dgrp_windows_onehot": (cint, [vp, i64, i64, i64, i64, i64, cint, vp, vp]),
    "dgrp_model_create": (cint, [C.POINTER(vp), cint, cint, cint, cint, vp, vp, vp, vp, vp, vp]),
    "dgrp_model_create_lstm": (cint, [C.POINTER(vp), cint, cint, cint, vp, vp, vp, vp, vp]),
    "dgrp_model_destroy": (cint, [vp]),
    "dgrp_model_dims": (cint, [vp, C.POINTER(cint), C.POINTER(cint), C.POINTER(cint), C.POINTER(cint)]),
    "dgrp_model_flags": (cint, [vp]),
    "dgrp_model_set_precision": (cint, [vp, cint]),
    "dgrp_model_view": (cint, [vp, cint, C.POINTER(vp)]),
    "dgrp_forward_workspace_bytes": (i64, [vp, i64]),
    "dgrp_forward_window_chunk": (i64, [vp]),
    "dgrp_forward_windows": (cint, [vp, vp, i64, i64, i64, i64, vp, vp, i64, vp]),
    "dgrp_forward_merge": (cint, [vp, vp, i64, i64, i64, i64, i64, vp, vp, i64, vp]),
    "dgrp_forward_merge_record": (cint, [vp, vp, i64, i64, i64, vp, vp, i64, vp]),
    "dgrp_forward_merge_record_workspace_bytes": (i64, [vp, i64, i64]),
    "dgrp_forward_reference_workspace_bytes": (i64, [vp, i64]),
    "dgrp_forward_windows_reference": (cint, [vp, vp, i64, i64, i64, i64, vp, vp, i64, vp]),
    "dgrp_get_max": (cint, [vp, i64, vp, i64, i64, i64, i64, vp]),
    "dgrp_scores": (cint, [vp, i64, cint, vp, vp, vp]),
    "dgrp_softmax_labels": (cint, [vp, i64, cint, vp, vp, vp, i64, vp]),
    "dgrp_mss_workspace_bytes": (i64, [i64]),
    "dgrp_mss_labels": (cint, [vp, vp, i64, cint, cint, cint, vp, vp, vp, i64, vp]),
    "dgrp_mss_segments_host": (cint, [vp, i64, vp, i64, C.POINTER(i64)]),
    "dgrp_mss_batch_workspace_bytes": (i64, [i64, i64]),
    "dgrp_mss_labels_batch": (cint, [vp, vp, i64, i64, vp, cint, cint, cint, vp, vp, i64, vp]),
    "dgrp_segments_workspace_bytes": (i64, [i64]),
    "dgrp_segments": (cint, [vp, i64, i64, i32, vp, i64, vp, vp, i64, vp]),
    "dgrp_record_workspace_bytes": (i64, [vp, i64, i64, cint]),
    "dgrp_predict_record": (cint, [vp, vp, i64, i64, i64, cint, cint, cint, i64, i32, vp, i64, C.POINTER(i64), vp, i64, vp]),
    "dgrp_batch_workspace_bytes": (i64, [vp, i64, vp, i64]),
    "dgrp_predict_batch": (cint, [vp, vp, i64, vp, vp, vp, vp, i64, i64, cint, cint, vp, i64, C.POINTER(i64), vp, i64, vp]),
    "dgrp_confusion_matrix": (cint, [vp, vp, i64, cint, vp, vp, vp]),
    "dgrp_filter_segments": (cint, [vp, vp, i64, i64, vp]),
    "dgrp_kernel_timer_enable": (cint, [cint]),
    "dgrp_kernel_timer_read": (cint, [C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(i64)]),
}

_lib = None


def lib() -> C.CDLL:
    """The loaded library; raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C deepgrp_amd/csrc` (hipcc, gfx950) "
                "or `python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback.")
        # PyTorch-ROCm ships its own libamdhip64; it must be in the process BEFORE this library is
        # loaded so that both bind to the same HIP runtime (streams and device pointers are shared).
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        if L.dgrp_abi_version() != 1:
            raise ImportError("libdeepgrp_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().dgrp_last_error().decode("utf-8", "replace")
        raise DgrpError(f"{what or 'libdeepgrp_hip'} failed (code {rc}): {msg}")


def exported_symbols():
    return sorted(_SIGNATURES)
