"""N2 on the GPU: dgrp_confusion_matrix / dgrp_filter_segments and the prediction.py mirror functions
built on them, against the oracle (bit-exact: integer work) and at BASELINE size through properties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    from deepgrp_amd.pipeline import require_gpu
    return require_gpu()


@pytest.mark.parametrize("n,ncls", [(1, 2), (15, 3), (16, 5), (17, 5), (4097, 4), (1_000_003, 5), (300_000, 16)])
def test_confusion_matrix_vs_oracle(dev, orc, n, ncls):
    import torch
    from deepgrp_amd import prediction as dgpred
    rng = np.random.default_rng(n)
    t = rng.integers(0, ncls, n)
    p = np.where(rng.random(n) < 0.7, t, rng.integers(0, ncls, n))
    want = orc.confusion_matrix(t, p)
    got = dgpred.confusion_matrix(t, p)
    np.testing.assert_array_equal(got, want)
    # int8 device tensors go straight to the kernel
    got2 = dgpred.confusion_matrix(torch.from_numpy(t.astype(np.int8)).to(dev), torch.from_numpy(p.astype(np.int8)).to(dev))
    np.testing.assert_array_equal(got2, want)


def test_confusion_matrix_reference_quirks(dev, orc):
    from deepgrp_amd import prediction as dgpred
    with pytest.raises(IndexError):
        dgpred.confusion_matrix(np.array([2, 3, 4]), np.array([2, 2, 4]))
    t, p = np.array([-1, 0, 1, -1]), np.array([0, 0, 1, -1])
    np.testing.assert_array_equal(dgpred.confusion_matrix(t, p), orc.confusion_matrix(t, p))


def test_calculate_metrics_vs_sklearn(dev):
    """The reference's test_calculate_metrics with scikit-learn in place of pycm."""
    from sklearn import metrics as skm
    from deepgrp_amd import prediction as dgpred
    rng = np.random.default_rng(11)
    truelbl = rng.choice([0, 1, 2, 3], size=100, replace=True)
    predlbl = rng.choice([0, 1, 2, 3], size=100, replace=True)
    cnf, stats = dgpred.calculate_metrics(predlbl, truelbl)
    np.testing.assert_array_equal(cnf, skm.confusion_matrix(truelbl, predlbl, labels=[0, 1, 2, 3]))
    np.testing.assert_allclose(stats["TotalACC"], skm.accuracy_score(truelbl, predlbl))
    np.testing.assert_allclose(stats["MCC"], skm.matthews_corrcoef(truelbl, predlbl))
    np.testing.assert_allclose(stats["F1"], skm.f1_score(truelbl, predlbl, average=None))


@pytest.mark.parametrize("min_len", (10, 20))
def test_filter_segments_reference_construction(dev, min_len):
    """tests/test_prediction.py:183-195 of the reference (float array, in place)."""
    from deepgrp_amd import prediction as dgpred
    segment_length = min_len * 2
    data = np.zeros(1000)
    data[110:110 + segment_length] = 1
    data[210 + segment_length:210 + 2 * segment_length] = 1
    expected = data.copy()
    data[0:min_len - 1] = 1
    data[120 + segment_length:120 + segment_length + min_len - 1] = 1
    data[(-min_len) + 1:] = 1
    dgpred.filter_segments(data, min_len=min_len)
    np.testing.assert_equal(data, expected)


@pytest.mark.parametrize("n,min_len,ncls", [(1, 1, 3), (1, 2, 3), (64, 3, 2), (5000, 50, 5), (5000, 1, 5), (300_000, 7, 16),
                                            (2_000_000, 50, 5)])
def test_filter_segments_vs_oracle(dev, orc, n, min_len, ncls):
    import torch
    from deepgrp_amd import prediction as dgpred
    rng = np.random.default_rng(n + min_len)
    runs = rng.geometric(1.0 / max(2, min_len), size=n)
    a = np.repeat(rng.integers(0, ncls, size=n), runs)[:n]
    want = orc.filter_segments(a, min_len)
    host = a.copy()
    dgpred.filter_segments(host, min_len)                    # numpy array, written back in place
    np.testing.assert_array_equal(host, want)
    d = torch.from_numpy(a.astype(np.int8)).to(dev)
    dgpred.filter_segments(d, min_len)                       # int8 device tensor, in place on the device
    np.testing.assert_array_equal(d.cpu().numpy(), want)


def test_filter_and_confusion_at_baseline_size(dev):
    """50 Mbp of run-structured labels: idempotence, no positive run shorter than min_len survives, bases of long
    runs and non-positive bases are untouched; the confusion matrix sums to n and its trace counts the agreements."""
    import torch
    from deepgrp_amd import prediction as dgpred
    n, min_len = 50_000_000, 50
    g = torch.Generator(device="cpu").manual_seed(5)
    runs = torch.randint(1, 120, (n // 40,), generator=g)
    vals = torch.randint(0, 5, (n // 40,), generator=g).to(torch.int8)
    a = torch.repeat_interleave(vals, runs)[:n].contiguous()
    n = a.numel()
    d = a.to(dev)
    before = d.clone()
    dgpred.filter_segments(d, min_len)
    again = d.clone()
    dgpred.filter_segments(again, min_len)
    assert torch.equal(d, again)
    changed = d != before
    assert bool((d[changed] == 0).all()) and bool((before[changed] > 0).all())
    # run lengths of the result: every positive run >= min_len
    h = d.cpu().numpy()
    edges = np.flatnonzero(np.diff(h) != 0) + 1
    starts = np.concatenate([[0], edges]); ends = np.concatenate([edges, [n]])
    pos = h[starts] > 0
    # a surviving positive run may have grown?  no: clearing only creates zeros, so it is an original run
    assert ((ends - starts)[pos] >= min_len).all()
    hb = before.cpu().numpy()
    eb = np.flatnonzero(np.diff(hb) != 0) + 1
    sb = np.concatenate([[0], eb]); nb = np.concatenate([eb, [n]])
    keep = (hb[sb] <= 0) | (nb - sb >= min_len)
    mask = np.repeat(keep, nb - sb)
    np.testing.assert_array_equal(h[mask], hb[mask])
    assert (h[~mask] == 0).all()
    cnf = dgpred.confusion_matrix(before, d)
    assert cnf.sum() == n and np.trace(cnf) == int((before == d).sum())
    assert (cnf[0, 1:] == 0).all()                           # nothing is ever created


@pytest.mark.parametrize("use_mss", (True, False))
@pytest.mark.parametrize("attention", (False, True))
def test_predict_complete(dev, orc, tmp_path, use_mss, attention):
    """prediction.py:114-141 end to end from a Keras HDF5 file in `logdir`, against the oracle's statement of the
    same chain (labels compared: the probabilities differ within the forward tolerance)."""
    from deepgrp_amd import model as dgmodel, prediction as dgpred
    from deepgrp_amd.preprocessing import Data
    u, T, C_, s, B = 32, 40, 5, 10, 7
    w = orc.Weights.random(u, C_, T, attention, seed=3, gain=1.0)
    dgmodel.save_keras_hdf5(str(tmp_path / "model.hdf5"), w.kernel, w.recurrent, w.bias, w.ff_kernel, w.ff_bias, w.scale, vecsize=T)
    rng = np.random.default_rng(9)
    n = 1003
    idx = rng.integers(0, 4, n).astype(np.uint8)
    fwd = np.zeros((5, n), np.int8)
    fwd[idx, np.arange(n)] = 1
    data = Data(fwd, np.zeros((C_, n), np.int8))
    opt = dgmodel.Options(vecsize=T, units=u, attention=attention, batch_size=B, min_mss_len=5, xdrop_len=5)
    got = dgpred.predict_complete(s, opt, tmp_path, data, use_mss=use_mss)
    merged = orc.predict_merged(idx, lambda w0, nw: orc.nn_forward(idx, w, s, w0, nw, np.float32), T, C_, s, B)
    assert got.shape == (n, C_)
    if use_mss:
        want = orc.labels_from_merged(merged, 5, 5, True)
        agree = (got.argmax(axis=1) == want).mean()
        assert got.dtype == np.float64 and agree > 0.995
    else:
        e = np.exp(merged - merged.max())
        np.testing.assert_allclose(got, e / e.sum(axis=1, keepdims=True), atol=1e-3)
    # options that disagree with the file are refused
    with pytest.raises(dgmodel.ModelFormatError):
        dgpred.predict_complete(s, dgmodel.Options(vecsize=T + 1, units=u, attention=attention), tmp_path, data)
    with pytest.raises(dgmodel.ModelFormatError):
        (tmp_path / "empty").mkdir(exist_ok=True)
        dgpred.setup_prediction_from_options_checkpoint(opt, tmp_path / "empty")
