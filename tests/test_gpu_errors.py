"""Error paths of the C ABI that need a model on the device: every misuse comes back as a code + message and leaves
the output buffers untouched; nothing is launched with operands that do not match the kernel's assumptions.
(The argument checks that need no device are in tests/test_abi_errors.py.)"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

EINVAL, ENOMEM = -1, -3


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    return torch.device("cuda")


@pytest.fixture(scope="module")
def L():
    from deepgrp_amd import _lib
    return _lib.lib()


def _sp():
    return torch.cuda.current_stream().cuda_stream


def _models(orc):
    from deepgrp_amd.pipeline import DeviceModel
    out = []
    for att in (False, True):
        w = orc.Weights.random(32, 5, 40, att, seed=3, gain=1.0)
        out.append(DeviceModel(w.kernel, w.recurrent, w.bias, w.ff_kernel, w.ff_bias, w.scale, vecsize=40))
    return out


def _err(L):
    return L.dgrp_last_error().decode()


def test_forward_shape_errors(L, dev, orc):
    plain, att = _models(orc)
    n = 1000
    idx = torch.randint(0, 5, (n,), dtype=torch.uint8, device=dev)
    total = L.dgrp_window_count(n, 40, 10)
    out = torch.full((total, 40, 5), 7.0, device=dev)
    merged = torch.full((n, 5), 7.0, device=dev)
    h = plain.handle
    # a window that would read past the record
    assert L.dgrp_forward_windows(h, idx.data_ptr(), n, 10, total - 1, 3, out.data_ptr(), None, 0, _sp()) == EINVAL
    assert "runs past" in _err(L)
    assert L.dgrp_forward_windows(h, idx.data_ptr(), n, 0, 0, 1, out.data_ptr(), None, 0, _sp()) == EINVAL     # stride 0
    assert L.dgrp_forward_windows(h, idx.data_ptr(), n, 10, -1, 1, out.data_ptr(), None, 0, _sp()) == EINVAL    # w0 < 0
    assert L.dgrp_forward_windows(h, None, n, 10, 0, 1, out.data_ptr(), None, 0, _sp()) == EINVAL
    assert "NULL" in _err(L)
    # merge: the reference's batch placement needs batch >= 1, and w0 + nw may not pass the record's window count
    assert L.dgrp_forward_merge(h, idx.data_ptr(), n, 10, 0, 0, total, merged.data_ptr(), None, 0, _sp()) == EINVAL
    assert "batch" in _err(L)
    assert L.dgrp_forward_merge(h, idx.data_ptr(), n, 10, 256, 1, total, merged.data_ptr(), None, 0, _sp()) == EINVAL
    # zero windows is a valid no-op
    assert L.dgrp_forward_windows(h, idx.data_ptr(), n, 10, 0, 0, out.data_ptr(), None, 0, _sp()) == 0
    # the attention model needs its spill workspace: without it DGRP_ENOMEM, with the message naming the size
    need = L.dgrp_forward_workspace_bytes(att.handle, total)
    assert need >= total * 40 * 32 * 2 and L.dgrp_forward_workspace_bytes(plain.handle, total) <= 256   # plain GRU: nominal
    assert L.dgrp_forward_windows(att.handle, idx.data_ptr(), n, 10, 0, total, out.data_ptr(), None, 0, _sp()) == ENOMEM
    assert str(need) in _err(L)
    short = torch.empty(need - 1, dtype=torch.uint8, device=dev)
    assert L.dgrp_forward_windows(att.handle, idx.data_ptr(), n, 10, 0, total, out.data_ptr(), short.data_ptr(), need - 1, _sp()) == ENOMEM
    torch.cuda.synchronize()
    assert bool((out == 7.0).all()) and bool((merged == 7.0).all())          # nothing was written by any refused call
    plain.close(), att.close()


def test_predict_record_and_batch_errors(L, dev, orc):
    from deepgrp_amd._lib import Segment
    plain, att = _models(orc)
    n = 5000
    idx = torch.randint(0, 4, (n,), dtype=torch.uint8, device=dev)
    cnt = C.c_int64(-5)
    recs = torch.zeros(64 * C.sizeof(Segment), dtype=torch.uint8, device=dev)
    need = L.dgrp_record_workspace_bytes(plain.handle, n, 10, 1)
    work = torch.empty(need, dtype=torch.uint8, device=dev)
    args = lambda wb, batch=256, s=10: (plain.handle, idx.data_ptr(), n, s, batch, 50, 50, 1, 0, 0, recs.data_ptr(), 64,
                                        C.byref(cnt), work.data_ptr(), wb, _sp())
    assert L.dgrp_predict_record(*args(need - 1)) == ENOMEM and str(need) in _err(L)
    assert cnt.value == 0                                                    # the count is defined even on failure
    assert L.dgrp_predict_record(*args(need, batch=0)) == EINVAL
    assert L.dgrp_predict_record(*args(need, s=0)) == EINVAL
    assert L.dgrp_predict_record(*args(need)) == 0 and cnt.value >= 0
    # n == 0: no device work, count 0
    cnt.value = 9
    assert L.dgrp_predict_record(plain.handle, None, 0, 10, 256, 50, 50, 1, 0, 0, None, 0, C.byref(cnt), None, 0, _sp()) == 0
    assert cnt.value == 0

    # batch: empty records do not belong in a batch, short workspace, NULL tables
    h_off = np.array([0, 2000], np.int64)
    h_n = np.array([2000, 0], np.int64)
    h_sp = np.zeros(2, np.int64)
    h_c = np.zeros(2, np.int32)
    p = lambda a: a.ctypes.data

    def batch(nn, wb, work_t):
        return L.dgrp_predict_batch(plain.handle, idx.data_ptr(), 2, p(h_off), p(nn), p(h_sp), p(h_c), 10, 256, 50, 50,
                                    recs.data_ptr(), 64, C.byref(cnt), work_t.data_ptr(), wb, _sp())

    assert batch(h_n, need, work) == EINVAL and "empty records" in _err(L)
    good_n = np.array([2000, 3000], np.int64)
    bneed = L.dgrp_batch_workspace_bytes(plain.handle, 2, p(good_n), 10)
    bwork = torch.empty(bneed, dtype=torch.uint8, device=dev)
    assert batch(good_n, bneed - 1, bwork) == ENOMEM and str(bneed) in _err(L)
    assert batch(good_n, bneed, bwork) == 0
    assert L.dgrp_predict_batch(plain.handle, idx.data_ptr(), 2, None, p(good_n), p(h_sp), p(h_c), 10, 256, 50, 50,
                                recs.data_ptr(), 64, C.byref(cnt), bwork.data_ptr(), bneed, _sp()) == EINVAL
    # zero records is a valid no-op
    cnt.value = 9
    assert L.dgrp_predict_batch(plain.handle, None, 0, None, None, None, None, 10, 256, 50, 50, None, 0, C.byref(cnt),
                                None, 0, _sp()) == 0 and cnt.value == 0
    plain.close(), att.close()


def test_model_create_rejects_bad_shapes_on_device(L, orc):
    """dgrp_model_create validates the shape before it copies anything; *out stays NULL on failure."""
    w = orc.Weights.random(32, 5, 40, False, seed=3, gain=1.0)
    arrs = [np.ascontiguousarray(a, np.float32) for a in (w.kernel, w.recurrent, w.bias, w.ff_kernel, w.ff_bias)]
    k, r, b, fk, fb = [a.ctypes.data for a in arrs]
    ptrs = [k, r, b, None, fk, fb]                                           # no attention scale
    h = C.c_void_p(123)
    assert L.dgrp_model_create(C.byref(h), 40, 3000, 5, 0, *ptrs) == EINVAL and h.value is None         # (up to 2048 units: the fp32 path)
    h = C.c_void_p(123)
    assert L.dgrp_model_create(C.byref(h), 40, 32, 5, 1, *ptrs) == EINVAL and h.value is None
    assert L.dgrp_model_create(C.byref(h), 40, 32, 5, 0, *ptrs) == 0 and h.value
    assert L.dgrp_model_destroy(h) == 0
    assert L.dgrp_model_destroy(None) == 0                                   # like free(NULL)


def test_python_mirror_raises_with_the_library_message(dev, orc):
    """The mirror package surfaces the same failures as exceptions carrying dgrp_last_error()."""
    from deepgrp_amd._lib import DgrpError
    plain, att = _models(orc)
    idx = torch.randint(0, 5, (300,), dtype=torch.uint8, device=dev)
    with pytest.raises(DgrpError, match="runs past"):
        plain.forward_windows(idx, 10, 20, 10)
    plain.close(), att.close()
