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


# ------------------------------------------------------------------------- scripts (N4)
def test_parse_rm_reference_fixture(tmp_path):
    """The reference's own fixture pair (tests/test_parse_rm_input.out -> tests/test_parse_rm_expect.bed, data files
    copied under tests/golden/, the input gzip-compressed), driven like its tests/test_parse_rm.py."""
    import gzip, os
    from conftest import GOLDEN
    from deepgrp_amd._scripts import parse_rm
    src = tmp_path / "input.out"
    src.write_bytes(gzip.open(os.path.join(GOLDEN, "parse_rm_input.out.gz"), "rb").read())
    out = tmp_path / "results.bed"
    parse_rm.main(["-o", str(out), str(src)])
    assert out.read_text() == open(os.path.join(GOLDEN, "parse_rm_expect.bed")).read()


def test_parse_rm_formats_and_hsat2(capsys, tmp_path):
    from deepgrp_amd._scripts import parse_rm
    ex, off = parse_rm.pentamer_sets()
    assert len(ex) == 10 and "GGAAT" in ex and "ATTCC" in ex and "AATGG" in ex
    lines = [
        "  693  29.9 10.1  6.3  chr21     9411195 9411890 (38718005) C  L1MEg          LINE/L1             (5558) 1775   1250 4456451\n",
        "   12  25.2  5.2  1.7  chr21     9412401 9412458 (38717437) +  (TATAT)n       Simple_repeat            1   60    (0) 4456452\n",
        "   50  10.0  0.0  0.0  chr1      101 200 (5) +  (GGAATGGAAC)n  Simple_repeat  1 100 (0) 7\n",        # exact + one-off
        "   50  10.0  0.0  0.0  chr1      101 200 (5) +  (GGAACGGAAC)n  Satellite  1 100 (0) 8\n",           # one-off only
        "   50  10.0  0.0  0.0  chr1      301 400 (5) +  (CATTC)n  Satellite  1 100 (0) 9\n",               # rotation of the rc
        "585\t463\t13\t6\t17\tchr2\t1000\t1300\t-500\t-\tAluSx\tSINE\tAlu\n",                                 # rmsk, class != family
        "585\t463\t13\t6\t17\tchr2\t2000\t2300\t-500\t+\tHSATII\tSatellite\tSatellite\n",                    # rmsk, by repeat name
        "585\t463\t13\t6\t17\tchr2\t3000\t3300\t-500\t+\tFoo\tDNA\tDNA\n",                                   # unnumbered
        "SW  perc perc perc  query  position in query\n",
        "\n",
    ]
    got = [str(r) for r in parse_rm.read_repeatmasker(ex, off, lines)]
    assert got == ["chr21\t9411194\t9411890\t4\tL1MEg\tLINE/L1",
                   "chr1\t100\t200\t1\t(GGAATGGAAC)n\tSimple_repeat",
                   "chr1\t300\t400\t1\t(CATTC)n\tSatellite",
                   "chr2\t1000\t1300\t3\tAluSx\tSINE/Alu",
                   "chr2\t2000\t2300\t1\tHSATII\tSatellite"]
    src = tmp_path / "in.out"
    src.write_text("".join(lines))
    parse_rm.main([str(src)])
    assert capsys.readouterr().out.splitlines() == got


def test_preprocess_sequence_script(tmp_path):
    """tests/test_preprocess_sequence.py of the reference (same input, expected array and md5), plus the rebuild
    rules and the KeyError on other letters."""
    import gzip, os
    from deepgrp_amd._scripts import preprocess_sequence
    from deepgrp_amd.preprocessing import load_onehot_npz
    out = tmp_path / "inputs.fa.gz.npz"
    src = tmp_path / "inputs.fa.gz"
    with gzip.open(src, "w") as fh:
        fh.write(">test\nACGTNACGTN\n".encode("utf-8"))
    preprocess_sequence.main([str(src)])
    got = np.load(out)
    expected = [[1, 0, 0, 0, 0], [0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 1]] * 2
    np.testing.assert_array_equal(got["fwd"], np.array(expected).T)
    assert got["fwd"].dtype == np.int8
    assert got["hash"][0] == "ff8ed7aaa145d49602bf5fdf5e5b8338"
    np.testing.assert_array_equal(load_onehot_npz(out), got["fwd"])
    stamp = os.path.getmtime(out)
    os.utime(out, (stamp - 100, stamp - 100))
    preprocess_sequence.main([str(src)])                         # unchanged hash: not rewritten
    assert os.path.getmtime(out) == stamp - 100
    preprocess_sequence.main([str(src), "--force"])
    assert os.path.getmtime(out) > stamp - 100
    with gzip.open(src, "w") as fh:
        fh.write(b">test\nacgt\nNN\n")                           # lower case is upper-cased, hash over the raw lines
    preprocess_sequence.main([str(src)])
    got = np.load(out)
    np.testing.assert_array_equal(got["fwd"].argmax(axis=0), [0, 1, 2, 3, 4, 4])
    with gzip.open(src, "w") as fh:
        fh.write(b">test\nACGR\n")
    with pytest.raises(KeyError):
        preprocess_sequence.main([str(src)])
    with pytest.raises(SystemExit):
        preprocess_sequence.main([str(tmp_path / "missing.fa.gz")])


def test_preprocess_y_reference_case(tmp_path):
    """The case of the reference's tests/test_preprocessing.py:13-31."""
    from deepgrp_amd.preprocessing import preprocess_y
    rows = [["chr1", 5, 10, 2, "X"], ["chr2", 6, 11, 5, "X"], ["chr1", 13, 15, 4, "X"], ["chr1", 16, 18, 7, "X"]]
    path = tmp_path / "data.bed"
    path.write_text("".join("\t".join(str(c) for c in r) + "\n" for r in rows))
    got = preprocess_y(filename=str(path), chromosom="chr1", length=20, repeats_to_search=[1, 2, 3, 4])
    expected = np.zeros((5, 20))
    expected[0, :5] = 1
    expected[2, 5:10] = 1
    expected[0, 10:13] = 1
    expected[4, 13:15] = 1
    expected[0, 15:] = 1
    np.testing.assert_equal(got, expected)


@pytest.mark.parametrize("start_n", (0, 10, 20))
@pytest.mark.parametrize("end_n", (0, 10, 20))
def test_drop_start_end_n_reference_case(start_n, end_n):
    """tests/test_preprocessing.py:34-53 of the reference, incl. the dropped last base."""
    from deepgrp_amd.preprocessing import drop_start_end_n
    testdata = np.zeros((5, 100))
    if end_n == 0:
        testdata[1, start_n:] = 1
    else:
        testdata[1, start_n:-end_n] = 1
        testdata[4, -end_n:] = 1
    testdata[4, :start_n] = 1
    truelbl = np.arange(100).reshape((1, 100))
    got_x, got_y = drop_start_end_n(testdata, truelbl)
    assert got_y.shape == (1, 100 - start_n - end_n - 1)
    assert got_x.shape == (5, 100 - start_n - end_n - 1)
    np.testing.assert_equal(got_x.sum(axis=1), [0, 100 - start_n - end_n - 1, 0, 0, 0])
    assert got_y[0, 0] == start_n
    assert got_y[0, -1] == 100 - end_n - 2
    np.testing.assert_equal(got_y[0, :-1] - got_y[0, 1:], -1)
