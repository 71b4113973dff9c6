"""N2 / N4 on the CPU: the oracle's restatement of confusion_matrix / filter_segments against
independent statements (scikit-learn, brute force, the reference's own test construction), the numpy metric
formulas of the mirror, and the preprocessing mirror against fixtures generated from the reference module
(oracle/make_golden_prep.py)."""
import numpy as np
import pytest

from conftest import golden


# ------------------------------------------------------------------------- confusion matrix / metrics
@pytest.mark.parametrize("ncls,n", [(4, 100), (2, 7), (5, 10000), (16, 3000)])
def test_oracle_confusion_matrix_vs_sklearn(orc, ncls, n):
    from sklearn.metrics import confusion_matrix as sk_cnf
    rng = np.random.default_rng(ncls * n)
    t = rng.integers(0, ncls, n)
    p = rng.integers(0, ncls, n)
    t[0], p[0] = 0, ncls - 1                 # make sure the label range is the full one
    got = orc.confusion_matrix(t, p)
    np.testing.assert_array_equal(got, sk_cnf(t, p, labels=list(range(ncls))))


def test_oracle_confusion_matrix_reference_quirks(orc):
    """prediction.py:216-221: the matrix has (max - min + 1) rows but is indexed with the raw labels."""
    # minimum above zero: labels 2..4 -> 3 x 3 matrix, label 3 is out of bounds
    with pytest.raises(IndexError):
        orc.confusion_matrix(np.array([2, 3, 4]), np.array([2, 2, 4]))
    # labels 1..2 -> 2 x 2, label 2 out of bounds
    with pytest.raises(IndexError):
        orc.confusion_matrix(np.array([1, 2]), np.array([1, 1]))
    # negative labels wrap like numpy indices: -1..1 -> 3 x 3, -1 lands in row/column 2
    got = orc.confusion_matrix(np.array([-1, 0, 1, -1]), np.array([0, 0, 1, -1]))
    want = np.zeros((3, 3), int)
    for i, j in zip([-1, 0, 1, -1], [0, 0, 1, -1]):
        want[i, j] += 1
    np.testing.assert_array_equal(got, want)
    with pytest.raises(ValueError):
        orc.confusion_matrix(np.array([], int), np.array([], int))


def test_metric_formulas_vs_sklearn():
    """_calculate_metrics / calculate_multiclass_matthews_cc (prediction.py:144-201); the reference's own test
    checks them against pycm (absent here), scikit-learn states the same quantities."""
    from sklearn import metrics as skm
    from deepgrp_amd.prediction import _calculate_metrics, calculate_multiclass_matthews_cc
    rng = np.random.default_rng(3)
    t = rng.choice([0, 1, 2, 3], size=100, replace=True)
    p = rng.choice([0, 1, 2, 3], size=100, replace=True)
    cnf = skm.confusion_matrix(t, p, labels=[0, 1, 2, 3])
    m = _calculate_metrics(cnf)
    np.testing.assert_allclose(m["TPR"], skm.recall_score(t, p, average=None, labels=[0, 1, 2, 3]))
    np.testing.assert_allclose(m["PPV"], skm.precision_score(t, p, average=None, labels=[0, 1, 2, 3]))
    np.testing.assert_allclose(m["F1"], skm.f1_score(t, p, average=None, labels=[0, 1, 2, 3]))
    np.testing.assert_allclose(m["MCC"], skm.matthews_corrcoef(t, p))
    np.testing.assert_allclose(calculate_multiclass_matthews_cc(cnf), skm.matthews_corrcoef(t, p))
    for c in range(4):                                                   # one-vs-rest statements
        tt, pp = (t == c), (p == c)
        tn, fp, fn, tp = skm.confusion_matrix(tt, pp, labels=[False, True]).ravel()
        np.testing.assert_allclose(m["TNR"][c], tn / (tn + fp))
        np.testing.assert_allclose(m["NPV"][c], tn / (tn + fn))
        np.testing.assert_allclose(m["FPR"][c], fp / (fp + tn))
        np.testing.assert_allclose(m["FNR"][c], fn / (tp + fn))
        np.testing.assert_allclose(m["FDR"][c], fp / (tp + fp))
        np.testing.assert_allclose(m["ACC"][c], (tp + tn) / 100)
    assert set(m) == {"TPR", "TNR", "PPV", "NPV", "FPR", "FNR", "FDR", "ACC", "F1", "MCC"}


# ------------------------------------------------------------------------- filter_segments
@pytest.mark.parametrize("min_len", (10, 20))
def test_oracle_filter_segments_reference_construction(orc, min_len):
    """The array of the reference's tests/test_prediction.py:183-195."""
    segment_length = min_len * 2
    data = np.zeros(1000)
    data[110:110 + segment_length] = 1
    data[210 + segment_length:210 + 2 * segment_length] = 1
    expected = data.copy()
    data[0:min_len - 1] = 1
    data[120 + segment_length:120 + segment_length + min_len - 1] = 1
    data[(-min_len) + 1:] = 1
    np.testing.assert_equal(orc.filter_segments(data, min_len=min_len), expected)


def _filter_brute(a, min_len):
    a = np.asarray(a).copy()
    out = a.copy()
    i = 0
    while i < a.size:
        j = i
        while j < a.size and a[j] == a[i]:
            j += 1
        if a[i] > 0 and j - i < min_len:
            out[i:j] = 0
        i = j
    return out


@pytest.mark.parametrize("n,min_len,ncls", [(1, 1, 3), (1, 2, 3), (64, 3, 2), (5000, 50, 5), (5000, 1, 5), (3000, 7, 16)])
def test_oracle_filter_segments_vs_brute_force(orc, n, min_len, ncls):
    rng = np.random.default_rng(n + min_len)
    runs = rng.geometric(1.0 / max(2, min_len), size=n)
    vals = rng.integers(0, ncls, size=n)
    a = np.repeat(vals, runs)[:n]
    np.testing.assert_array_equal(orc.filter_segments(a, min_len), _filter_brute(a, min_len))
    np.testing.assert_array_equal(orc.filter_segments(-a, min_len), -a)       # non-positive labels are never touched


# ------------------------------------------------------------------------- preprocessing (N4)
def test_preprocess_y_golden(tmp_path):
    from deepgrp_amd.preprocessing import preprocess_y
    g = golden("preprocessing.npz")
    path = tmp_path / "rm.bed"
    path.write_bytes(g["bed_text"].tobytes())
    length = int(g["length"])
    k = 0
    while f"y{k}" in g:
        chrom, reps, err = str(g[f"y{k}_chrom"]), [int(r) for r in g[f"y{k}_reps"]], str(g[f"y{k}_err"])
        if err:
            with pytest.raises(Exception) as ei:
                preprocess_y(path, chrom, length, reps)
            assert type(ei.value).__name__ == err
        else:
            got = preprocess_y(path, chrom, length, reps)
            assert got.dtype == np.int8
            np.testing.assert_array_equal(got, g[f"y{k}"])
        k += 1
    assert k == 4


def test_drop_start_end_n_golden():
    from deepgrp_amd.preprocessing import drop_start_end_n
    g = golden("preprocessing.npz")
    k = 0
    while f"d{k}_fwd" in g:
        f2, l2 = drop_start_end_n(g[f"d{k}_fwd"], g[f"d{k}_lab"])
        np.testing.assert_array_equal(f2, g[f"d{k}_fwd_out"])
        np.testing.assert_array_equal(l2, g[f"d{k}_lab_out"])
        k += 1
    assert k == 5


def test_load_onehot_npz_roundtrip(tmp_path):
    """The file layout of preprocess_sequence.py:71-78: np.savez_compressed(<fasta.gz>, fwd=int8 [5, N], hash=[md5])."""
    from deepgrp_amd.preprocessing import Data, load_onehot_npz
    from deepgrp_amd import sequence as dgsequence
    import hashlib
    seq = "NNACGTNACGTTTGACNN"
    enc = np.zeros((5, len(seq)), np.int8)
    enc[["ACGTN".index(c) for c in seq], np.arange(len(seq))] = 1
    np.savez_compressed(tmp_path / "x.fa.gz", fwd=enc, hash=np.array([hashlib.md5(seq.encode()).hexdigest()]))
    fwd = load_onehot_npz(tmp_path / "x.fa.gz.npz")
    np.testing.assert_array_equal(fwd, enc)
    d = Data(fwd, np.zeros((5, len(seq)), np.int8))
    assert d.fwd is fwd and d.truelbl.shape == (5, len(seq))
    np.savez_compressed(tmp_path / "bad", fwd=np.zeros((4, 3), np.int8))
    with pytest.raises(ValueError):
        load_onehot_npz(tmp_path / "bad.npz")
    del dgsequence
