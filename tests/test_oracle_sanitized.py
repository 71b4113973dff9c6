"""The C restatement under AddressSanitizer + UBSan (CPU build only; SURVEY section 5 asks for it):
the golden MSS / score / segment cases and a small NN forward run through liboracle_asan.so in a
child interpreter with the ASan runtime preloaded."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CHILD = r"""
import ctypes as C, numpy as np, os, sys
sys.path.insert(0, os.environ["DGRP_ROOT"])
from oracle import oracle as orc
orc._LIB = None
_real = C.CDLL
def _cdll(path, *a, **k):
    return _real(path.replace("liboracle.so", "liboracle_asan.so"), *a, **k)
C.CDLL = _cdll
orc.C.CDLL = _cdll
L = orc.lib()
g = np.load(os.path.join(os.environ["DGRP_ROOT"], "tests", "golden", "probs_to_rows.npz"))
for k in range(int(g["count"])):
    probs = g[f"probs{k}"]; ml, xd, off = (int(v) for v in g[f"par{k}"])
    sc, cl = orc.scores(probs)
    lab = orc.find_mss_labels(sc, cl, probs.shape[1], ml, xd)
    assert np.array_equal(lab, g[f"labels{k}"])
    assert np.array_equal(orc.segments(lab, off), g[f"rows{k}"])
g = np.load(os.path.join(os.environ["DGRP_ROOT"], "tests", "golden", "mss_raw.npz"))
for k in range(int(g["count"])):
    nof, ml, xd = (int(v) for v in g[f"p{k}"])
    assert np.array_equal(orc.find_mss_labels(g[f"s{k}"], g[f"l{k}"], nof, ml, xd), g[f"o{k}"])
w = orc.Weights.random(24, 5, 30, True, seed=1)
idx = np.random.default_rng(0).integers(0, 5, size=400).astype(np.uint8)
a = orc.nn_forward(idx, w, 7, 0, 20, np.float64, threads=1)
assert abs(a.sum(axis=2) - 1).max() < 1e-9
lw = orc.LSTMWeights.random(16, 5, 30)
assert abs(orc.lstm_forward(idx, lw, 7, 0, 20, np.float32, threads=1).sum(axis=2) - 1).max() < 1e-5
st, oh = orc.one_hot_encode_dna_sequence("NNACGTNNacgtXN")
assert st == 2 and oh.shape == (5, 11)
merged = orc.merge_all(np.random.default_rng(1).random((17, 20, 5), dtype=np.float32), 20 + 16 * 3, 3, 4)
assert merged.shape == (68, 5)
rng = np.random.default_rng(2)
t, q = rng.integers(0, 5, size=5000), rng.integers(0, 5, size=5000)
cnf = orc.confusion_matrix(t, q)
assert cnf.sum() == 5000 and cnf[2, 3] == int(((t == 2) & (q == 3)).sum())
lab = np.repeat(rng.integers(0, 3, size=400), rng.integers(1, 9, size=400))
f = orc.filter_segments(lab, 5)
assert f.shape == lab.shape and (f[lab == 0] == 0).all()
assert orc.filter_segments(np.zeros(0, np.int64), 5).size == 0
print("sanitized run ok")
"""


def test_oracle_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not libasan or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"], check=True, capture_output=True)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1",
               DGRP_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "sanitized run ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
