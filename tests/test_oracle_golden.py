"""The oracle (oracle/dgrp_oracle.c) against golden vectors produced by the
reference's own compiled Cython/C (oracle/make_golden.py) -- this is what pins
the checker before any GPU result is compared with it."""
import numpy as np
import pytest

from conftest import golden


def test_mss_kat(orc):
    """tests/test_mss.py:10-24 of the reference: the 14-element known answer."""
    g = golden("mss_kat.npz")
    for ml in (0, 3, 10):
        for xd in (-1, 0, 10):
            got = orc.find_mss_labels(g["scores"], g["labels"], 3, ml, xd)
            np.testing.assert_array_equal(got, g[f"out_{ml}_{xd}"])
            expected = g["labels"].copy()
            if ml == 0:
                expected[2] = 1
                expected[11] = 1
            np.testing.assert_array_equal(got, expected)


def test_mss_raw(orc):
    g = golden("mss_raw.npz")
    for k in range(int(g["count"])):
        nof, ml, xd = (int(v) for v in g[f"p{k}"])
        got = orc.find_mss_labels(g[f"s{k}"], g[f"l{k}"], nof, ml, xd)
        np.testing.assert_array_equal(got, g[f"o{k}"], err_msg=f"case {k}")


def test_mss_segments_vs_compiled_reference_c(orc):
    """orc_mss_find_all against the reference's mss.c (oracle/_ref) on random folds,
    including scores/endpoints bit for bit."""
    if orc.ref_c() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    rng = np.random.default_rng(5)
    for n in (0, 1, 5, 100, 3000, 50000):
        for trial in range(6):
            S = rng.normal(-0.2, 2.0, size=n) if trial % 2 else rng.integers(-5, 4, size=n).astype(float)
            for (msc, xd) in ((0.0, -1.0), (3.7, 2.5), (229.756, 2297.56), (1.0, 0.5)):
                a = orc.mss_find_all(S, msc, xd)
                b = orc.mss_find_all(S, msc, xd, use_ref=True)
                assert a == b


def test_probs_to_rows(orc):
    """prediction.py:51-59 -> pymss.pyx -> sequence.pyx:79-85: scores, classes, labels
    and TSV rows all bit-identical to the reference run in the build container."""
    g = golden("probs_to_rows.npz")
    for k in range(int(g["count"])):
        probs = g[f"probs{k}"]
        ml, xd, off = (int(v) for v in g[f"par{k}"])
        sc, cl = orc.scores(probs)
        np.testing.assert_array_equal(cl, g[f"cls{k}"])
        np.testing.assert_array_equal(sc.view(np.int64), g[f"scores{k}"].view(np.int64))
        lab = orc.find_mss_labels(sc, cl, probs.shape[1], ml, xd)
        np.testing.assert_array_equal(lab, g[f"labels{k}"])
        np.testing.assert_array_equal(orc.segments(lab, off), g[f"rows{k}"])


def test_softmax_path(orc):
    g = golden("softmax.npz")
    sm, lab = orc.softmax_argmax(g["probs"])
    np.testing.assert_array_equal(sm.view(np.int32), g["softmax"].view(np.int32))
    np.testing.assert_array_equal(lab, g["labels"])


def test_onehot(orc):
    g = golden("onehot.npz")
    for i in range(int(g["count"])):
        s = bytes(g[f"seq{i}"]).decode()
        st, oh = orc.one_hot_encode_dna_sequence(s)
        assert st == int(g[f"start{i}"])
        np.testing.assert_array_equal(oh, g[f"onehot{i}"])
        st2, n = orc.strip_n(s.encode())
        np.testing.assert_array_equal(orc.encode_idx(s.encode()[st2:st2 + n]), oh.argmax(axis=0) if n else [])


def test_onehot_reference_test_semantics(orc):
    """tests/test_sequence.py:10-27 of the reference, replayed on the oracle."""
    rng = np.random.default_rng(0)
    data = "NNNN" + "".join(rng.choice(["A", "C", "G", "T", "N"], size=100)) + "NNNN"
    startpos, oh = orc.one_hot_encode_dna_sequence(data)
    np.testing.assert_equal(oh.sum(axis=0), 1)
    expected = data.translate(str.maketrans({"A": "0", "C": "1", "G": "2", "T": "3", "N": "4"})).strip("4")
    np.testing.assert_equal(oh.argmax(axis=0), np.array(list(expected)).astype(int))
    assert all(c == "N" for c in data[:startpos]) and data[startpos] != "N"
    with pytest.raises(ValueError):
        orc.one_hot_encode_dna_sequence("NNNN")


def test_get_max(orc):
    g = golden("get_max.npz")
    for k in range(int(g["count"])):
        out = g[f"init{k}"].copy()
        orc.get_max(out, g[f"in{k}"], int(g[f"stride{k}"]))
        np.testing.assert_array_equal(out, g[f"out{k}"])


@pytest.mark.parametrize("stride", [1, 2, 3])
def test_get_max_reference_test(orc, stride):
    """tests/test_sequence.py:47-56 of the reference."""
    x = np.zeros((10, 100, 5), np.float32)
    x[:, 0, :] = 1.0
    out = np.zeros((10000, 5), np.float32)
    got = orc.get_max(out, x, stride)
    for i in range(0, stride * 10, stride):
        np.testing.assert_equal(got[i], 1)
        got[i] -= 1.0
    np.testing.assert_equal(got, 0)


def test_placement(orc):
    """prediction.py:103-110 incl. the partial-last-batch offset."""
    g = golden("placement.npz")
    for k in range(int(g["count"])):
        N, T, s, B = (int(v) for v in g[f"par{k}"])
        probs = g[f"probs{k}"]
        assert probs.shape[0] == orc.window_count(N, T, s)
        np.testing.assert_array_equal(orc.merge_all(probs, N, s, B), g[f"merged{k}"])
        idx = np.zeros(N, np.uint8)
        drv = orc.predict_merged(idx, lambda w, b: probs[w:w + b], T, 5, s, B)
        np.testing.assert_array_equal(drv, g[f"merged{k}"])


def test_window_enumeration(orc):
    """tests/test_prediction.py:16-36 of the reference: ceil((N-T)/s) windows, contents."""
    rng = np.random.default_rng(1)
    idx = rng.integers(0, 5, size=200).astype(np.uint8)
    for step in (2, 4):
        for vec in (20, 30):
            total = int(np.ceil((200 - vec) / step))
            assert orc.window_count(200, vec, step) == total
            w = orc.windows_f32(idx, vec, step, 0, total)
            for i in range(total):
                np.testing.assert_array_equal(w[i].argmax(axis=1), idx[i * step:i * step + vec])
                np.testing.assert_array_equal(w[i].sum(axis=1), 1)
    assert orc.window_count(1000, 200, 50) == 16      # start 800 excluded
    assert orc.window_count(200, 200, 50) == 0 and orc.window_count(201, 200, 50) == 1


def test_segments(orc):
    g = golden("segments.npz")
    for k in range(int(g["count"])):
        lab = g[f"lab{k}"]
        allseg = g[f"all{k}"]
        np.testing.assert_array_equal(orc.segments(lab, 5), allseg[allseg[:, 2] > 0])
        # walk get_segments like yield_segments does
        i, walked = 0, []
        while i < lab.size:
            st, en, lb = orc.get_segments(lab, i)
            i = en
            walked.append((st + 5, en + 5, lb))
        np.testing.assert_array_equal(np.array(walked, np.int64).reshape(-1, 3), allseg)


@pytest.mark.parametrize("begin", (3, 5, 50))
@pytest.mark.parametrize("startpos", [0, 10, 22, 33])
@pytest.mark.parametrize("endpos", [0, 44, 54, 62])
@pytest.mark.parametrize("label", [1, 2, 3])
def test_get_segments_reference_test(orc, begin, startpos, endpos, label):
    """tests/test_sequence.py:30-44 of the reference."""
    data = np.zeros(100, dtype=np.int64)
    if endpos > 0:
        data[startpos:endpos] = label
    st, en, lb = orc.get_segments(data, begin)
    assert st == (max(startpos, begin) if endpos > begin else 99)
    assert en == (endpos if endpos > begin else 100)
    assert lb == (label if endpos > begin else 0)


def test_numpy_math(orc):
    """orc_np_logf / orc_np_expf are numpy's float32 routines bit for bit."""
    import ctypes
    L = orc.lib()
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.random(20000, dtype=np.float32) * 100,
                        np.exp(rng.normal(size=20000) * 3).astype(np.float32)])
    x = x[x > 0]
    got = np.array([L.orc_np_logf(ctypes.c_float(float(v))) for v in x], np.float32)
    np.testing.assert_array_equal(got.view(np.int32), np.log(x).view(np.int32))
    y = np.concatenate([-rng.random(20000, dtype=np.float32), (rng.normal(size=20000) * 5).astype(np.float32)])
    got = np.array([L.orc_np_expf(ctypes.c_float(float(v))) for v in y], np.float32)
    np.testing.assert_array_equal(got.view(np.int32), np.exp(y).view(np.int32))


def test_fasta_reader(orc):
    """__main__.py:20-43 semantics incl. dropped header-less prefix."""
    txt = ["ACGT\n", ">chr1 desc\n", "acgt\n", "NNa\n", ">chr2\n", ">chr3\n", "tt\n"]
    assert orc.read_multi_fasta(txt) == [("chr1 desc", "ACGTNNA"), ("chr2", ""), ("chr3", "TT")]
    with pytest.raises(IndexError):
        orc.read_multi_fasta([">a\n", "\n"])
