"""Batched entry points for files of many short records (dgrp_*_batch): every record of the batch must come out
exactly as from the per-record path / the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from deepgrp_amd.pipeline import require_gpu
    return require_gpu()


def _scores(rng, n, style):
    if style == "runs":                         # confident calls in runs, like a trained model
        lab = np.resize(np.repeat(rng.integers(0, 5, size=n // 40 + 2), rng.integers(1, 120, size=n // 40 + 2)), n)
        m = np.clip(rng.uniform(0.4, 0.999, n), None, 0.99).astype(np.float32)
        t = np.abs(np.log(m / (1 - m)))
        return np.where(lab > 0, t, -10 * t).astype(np.float64), lab.astype(np.int64)
    if style == "noise":
        return rng.normal(0, 5, n), rng.integers(0, 5, n).astype(np.int64)
    s = rng.integers(-6, 7, n).astype(np.float64) * 0.5   # small integers: zeros, ties, exact sums
    return s, rng.integers(0, 3, n).astype(np.int64)


@pytest.mark.parametrize("style", ("runs", "noise", "ints"))
@pytest.mark.parametrize("ml,xd", [(50, 50), (3, 10), (10, 0), (0, -1)])
def test_mss_labels_batch_vs_oracle(dev, orc, style, ml, xd):
    import torch
    from deepgrp_amd._lib import check, lib
    from deepgrp_amd.pipeline import stream_ptr
    L = lib()
    rng = np.random.default_rng(hash((style, ml, xd)) % 2**32)
    lens = [1, 2, 63, 64, 65, 128, 129, 1000, 4097, 20000, 7, 300] + [int(x) for x in rng.integers(1, 3000, 40)]
    starts = np.zeros(len(lens) + 1, np.int64)
    for i, n in enumerate(lens):
        starts[i + 1] = starts[i] + (n + 63) // 64 * 64
    total = int(starts[-1])
    S = np.zeros(total, np.float64)
    cls = np.zeros(total, np.int8)
    want = np.zeros(total, np.int8)
    for i, n in enumerate(lens):
        s, lab = _scores(rng, n, style)
        a = int(starts[i])
        S[a:a + n] = s
        cls[a:a + n] = lab
        want[a:a + n] = orc.find_mss_labels(s, lab, 5, ml, xd)
    d_S, d_cls = torch.from_numpy(S).to(dev), torch.from_numpy(cls).to(dev)
    d_out = torch.empty(total, dtype=torch.int8, device=dev)
    wb = L.dgrp_mss_batch_workspace_bytes(total, len(lens))
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    check(L.dgrp_mss_labels_batch(d_S.data_ptr(), d_cls.data_ptr(), total, len(lens), starts.ctypes.data, 5, ml, xd,
                                  d_out.data_ptr(), work.data_ptr(), wb, stream_ptr()), "dgrp_mss_labels_batch")
    got = d_out.cpu().numpy()
    for i, n in enumerate(lens):
        a = int(starts[i])
        np.testing.assert_array_equal(got[a:a + n], want[a:a + n], err_msg=f"record {i} n={n}")


@pytest.mark.parametrize("u,T,s,B,att", [(128, 200, 50, 256, False), (64, 40, 7, 9, False), (32, 30, 4, 7, False), (160, 50, 10, 16, False),
                                         (60, 342, 50, 256, True), (64, 40, 7, 9, True), (128, 30, 4, 7, True), (16, 70, 3, 5, True)])
def test_predict_batch_equals_record_by_record(dev, orc, u, T, s, B, att):
    """dgrp_predict_batch (one GRU launch, batched post-processing) against dgrp_predict_record per record: identical
    segment rows -- incl. records shorter than a window, of exactly one window, 64-aligned lengths and the
    partial-batch placement."""
    import torch
    from deepgrp_amd.pipeline import ContigPipeline, DeviceModel
    w = orc.Weights.random(u, 5, T, att, seed=u, gain=3.0)
    m = DeviceModel(w.kernel, w.recurrent, w.bias, w.ff_kernel, w.ff_bias, w.scale, vecsize=T)
    rng = np.random.default_rng(u + T)
    lens = [1, 2, T - 1, T, T + 1, 64, 128, T + s, T + 16 * s, 3 * T + 7, 4097] + [int(x) for x in rng.integers(1, 6000, 30)]
    gaps = rng.integers(0, 37, len(lens))                     # records sit at arbitrary byte offsets of the buffer
    offs, pos = [], 0
    for n, g in zip(lens, gaps):
        pos += int(g)
        offs.append(pos)
        pos += n
    base = rng.choice(5, size=pos + 5, p=[.24, .25, .25, .24, .02]).astype(np.uint8)
    d_base = torch.from_numpy(base).to(dev)
    pipe = ContigPipeline(m, s, B, 4, 6)
    assert pipe.batchable()
    sp = [int(x) for x in rng.integers(0, 1000, len(lens))]
    got = pipe.run_batch(d_base, offs, lens, sp, list(range(len(lens))))
    want = np.concatenate([pipe.run_idx(d_base[o:o + n].clone(), p0, contig=i) for i, (o, n, p0) in enumerate(zip(offs, lens, sp))])
    np.testing.assert_array_equal(got, want)
    assert len(want) > len(lens)
    m.close()


@pytest.mark.parametrize("u,T,s,B", [(48, 40, 7, 9), (128, 60, 13, 256), (100, 25, 3, 4)])
def test_predict_batch_lstm(dev, orc, u, T, s, B):
    """rnn = "LSTM" models through the batched path: identical rows to record by record."""
    import torch
    from deepgrp_amd.pipeline import ContigPipeline, DeviceModel
    w = orc.LSTMWeights.random(u, 5, T, seed=u, gain=2.0)
    m = DeviceModel(w.kernel, w.recurrent, w.bias, w.ff_kernel, w.ff_bias, None, vecsize=T, rnn="LSTM")
    rng = np.random.default_rng(u)
    lens = [1, T - 1, T, T + 1, 64, T + 16 * s, 2000] + [int(x) for x in rng.integers(1, 4000, 20)]
    offs, pos = [], 0
    for n in lens:
        pos += int(rng.integers(0, 20)); offs.append(pos); pos += n
    base = rng.choice(5, size=pos + 5, p=[.24, .25, .25, .24, .02]).astype(np.uint8)
    d_base = torch.from_numpy(base).to(dev)
    pipe = ContigPipeline(m, s, B, 4, 6)
    assert pipe.batchable()
    got = pipe.run_batch(d_base, offs, lens, [3] * len(lens), list(range(len(lens))))
    want = np.concatenate([pipe.run_idx(d_base[o:o + n].clone(), 3, contig=i) for i, (o, n) in enumerate(zip(offs, lens))])
    np.testing.assert_array_equal(got, want)
    m.close()

