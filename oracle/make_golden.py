#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (build container only).

Runs the reference's own compiled Cython/C (``make -C oracle ref`` builds them
from the sources where they lie under /root/reference into oracle/_ref/) on
seeded inputs and stores inputs + outputs as small fixtures.  Nothing of the
reference's source text is stored -- only data.

deepgrp/prediction.py cannot be imported here (it imports TensorFlow, absent
offline), so its four numpy-only functions are driven as follows: ``predict``'s
loop (prediction.py:103-110) is replayed with the compiled ``get_max`` and a
table of fake ``predict_on_batch`` outputs; ``apply_mss`` (:51-59) and
``softmax`` (:64-65) are evaluated with this image's numpy (2.2.6) expression
by expression and fed to the compiled ``find_mss_labels``.

Usage:  python oracle/make_golden.py      (writes tests/golden/)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, "/root/reference")                       # deepgrp.preprocessing for sequence.pyx:9
sys.path.insert(0, os.path.join(HERE, "_ref", "pyref"))
import sequence as refseq                                   # noqa: E402  compiled deepgrp/sequence.pyx
import mss as refmss                                        # noqa: E402  compiled deepgrp/_mss/pymss.pyx

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def ref_predict(probs_all, N, C, T, s, B):
    """prediction.py:103-110 with batches of a fake model output table."""
    nwin = probs_all.shape[0]
    predictions = np.zeros((N, C), dtype=np.float32)
    i = 0
    w = 0
    while w < nwin:
        batch = probs_all[w:w + B]
        index = i * batch.shape[0] * s
        refseq.get_max(predictions[index:], np.ascontiguousarray(batch), s)
        w += batch.shape[0]
        i += 1
    return predictions


def ref_apply_mss(probs, min_mss_len, xdrop_len):
    """prediction.py:51-59."""
    nof_labels = probs.shape[1]
    results_classes = probs.argmax(axis=1)
    mins = probs.max(axis=1) + 1e-6
    mins[mins > 0.99] = 0.99
    t_scores = np.log(mins / (1 - mins))
    assert t_scores.dtype == np.float32
    scores = np.where(results_classes > 0, t_scores, -10 * t_scores).astype(float)
    onehot = refmss.find_mss_labels(scores, results_classes, nof_labels, min_mss_len, xdrop_len)
    return scores, results_classes, onehot.argmax(axis=1)


def ref_softmax(array):
    """prediction.py:64-65."""
    e_x = np.exp(array - np.max(array))
    return e_x / e_x.sum(axis=1, keepdims=True)


def ref_rows(labels, startpos):
    """__main__.py:288-290."""
    return np.array([seg for seg in refseq.yield_segments(labels, startpos) if seg[2] > 0],
                    dtype=np.int64).reshape(-1, 3)


def run_structured_probs(rng, N, C, mean_run=300, conf=0.97, noise=0.25):
    """Per-base probabilities that look like a model's: long runs of one class,
    mostly confident, with uncertain flanks and a few exact ties / zeros."""
    labels = np.zeros(N, np.int64)
    i = 0
    while i < N:
        L = int(rng.geometric(1.0 / mean_run))
        lab = 0 if rng.random() < 0.6 else int(rng.integers(1, C))
        labels[i:i + L] = lab
        i += L
    p = rng.dirichlet(np.full(C, noise), size=N).astype(np.float32)
    strength = rng.beta(6, 1.2, size=N).astype(np.float32) * conf
    onehot = np.eye(C, dtype=np.float32)[labels]
    probs = (1 - strength)[:, None] * p + strength[:, None] * onehot
    return np.ascontiguousarray(probs.astype(np.float32))


def main():
    meta = {"numpy": np.__version__, "generator": "oracle/make_golden.py",
            "reference": "fhausmann/deepgrp v0.2.3 compiled from /root/reference"}
    rng = np.random.default_rng(20240)

    # ---- 1. test_mss.py KAT (tests/test_mss.py:10-24) x 9 parameter pairs --------------
    kat_s = np.array([1, 1, -1, 1, 1, -4, 1, 1, -10, 1, 1, -1, 1, 1], dtype=np.float64)
    kat_l = np.array([1, 1, 0, 1, 1, 0, 2, 2, 0, 1, 1, 0, 2, 2], dtype=np.int64)
    kat = {}
    for ml in (0, 3, 10):
        for xd in (-1, 0, 10):
            kat[f"{ml}_{xd}"] = refmss.find_mss_labels(kat_s, kat_l, 3, ml, xd).argmax(axis=1)
    np.savez_compressed(os.path.join(OUT, "mss_kat.npz"), scores=kat_s, labels=kat_l,
                        **{f"out_{k}": v for k, v in kat.items()})

    # ---- 2. find_mss_labels on raw score arrays (integer-ish, ties, zeros, resets) -----
    cases = {}
    k = 0
    for n in (1, 2, 3, 17, 200, 5000):
        for style in ("small_int", "gauss", "ties"):
            for (ml, xd) in ((50, 50), (0, -1), (10, 0), (3, 10), (1, 1)):
                if style == "small_int":
                    sc = rng.integers(-6, 4, size=n).astype(np.float64)
                elif style == "gauss":
                    sc = rng.normal(-0.3, 3.0, size=n)
                else:
                    sc = rng.choice(np.array([4.5951, -45.951, 0.0, 138.155, -0.5, 0.5]), size=n)
                lab = rng.integers(0, 4, size=n).astype(np.int64)
                out = refmss.find_mss_labels(sc, lab, 4, ml, xd).argmax(axis=1)
                cases[f"s{k}"] = sc
                cases[f"l{k}"] = lab
                cases[f"p{k}"] = np.array([4, ml, xd], np.int64)
                cases[f"o{k}"] = out
                k += 1
    cases["count"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "mss_raw.npz"), **cases)

    # ---- 3. probabilities -> scores -> labels -> rows (prediction.py:40-59 + __main__) -
    e2e = {}
    k = 0
    for (N, C) in ((1000, 5), (20000, 5), (3000, 3)):
        probs = run_structured_probs(rng, N, C)
        probs[5:9] = 0.0                               # uncovered rows
        probs[N - 37:] = 0.0                           # the uncovered tail (Q1)
        probs[100] = 0.2 if C == 5 else 1.0 / 3        # exact tie -> argmax first
        for (ml, xd) in ((50, 50), (0, -1), (10, 0), (3, 10)):
            sc, cl, lab = ref_apply_mss(probs, ml, xd)
            rows = ref_rows(lab, 11)
            e2e[f"probs{k}"] = probs
            e2e[f"par{k}"] = np.array([ml, xd, 11], np.int64)
            e2e[f"scores{k}"] = sc
            e2e[f"cls{k}"] = cl
            e2e[f"labels{k}"] = lab
            e2e[f"rows{k}"] = rows
            k += 1
    # noisy (adversarial) probabilities: many tiny segments, non-representable sums
    probs = rng.dirichlet(np.full(5, 0.3), size=4000).astype(np.float32)
    for (ml, xd) in ((50, 50), (2, 1)):
        sc, cl, lab = ref_apply_mss(probs, ml, xd)
        e2e[f"probs{k}"] = probs
        e2e[f"par{k}"] = np.array([ml, xd, 0], np.int64)
        e2e[f"scores{k}"] = sc
        e2e[f"cls{k}"] = cl
        e2e[f"labels{k}"] = lab
        e2e[f"rows{k}"] = ref_rows(lab, 0)
        k += 1
    e2e["count"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "probs_to_rows.npz"), **e2e)

    # ---- 4. softmax path (prediction.py:62-65, __main__.py:81-83) ----------------------
    probs = run_structured_probs(rng, 4000, 5)
    probs[7:11] = 0.0
    sm = ref_softmax(probs)
    np.savez_compressed(os.path.join(OUT, "softmax.npz"), probs=probs, softmax=sm,
                        labels=sm.argmax(axis=1))

    # ---- 5. one-hot (sequence.pyx:19-36, :55-58) ---------------------------------------
    strings = ["NNACGTNNacgtXN", "ACGT", "A", "NA", "AN", "NNNNACGTRYKMSWBDHVNacgtnNNNN",
               "nnACGTnn", "N" * 30 + "".join(rng.choice(list("ACGTN"), size=500)) + "N" * 12,
               "".join(rng.choice(list("ACGTNacgtnRYKM*-. "), size=300))]
    oh = {"count": np.array(len(strings))}
    for i, sq in enumerate(strings):
        st, arr = refseq.one_hot_encode_dna_sequence(sq)
        oh[f"seq{i}"] = np.frombuffer(sq.encode(), np.uint8)
        oh[f"start{i}"] = np.array(st)
        oh[f"onehot{i}"] = arr
    np.savez_compressed(os.path.join(OUT, "onehot.npz"), **oh)

    # ---- 6. get_max (maxcalc.c:10-24) and the predict loop incl. partial batches -------
    gm = {}
    k = 0
    for (b, T, C, s) in ((10, 100, 5, 1), (10, 100, 5, 2), (10, 100, 5, 3), (7, 40, 5, 50), (3, 20, 3, 7)):
        x = rng.random((b, T, C), dtype=np.float32)
        out = rng.random(((b - 1) * s + T + 13, C), dtype=np.float32) * 0.5
        gm[f"in{k}"] = x
        gm[f"init{k}"] = out.copy()
        gm[f"stride{k}"] = np.array(s)
        gm[f"out{k}"] = refseq.get_max(out, x, s).copy()
        k += 1
    gm["count"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "get_max.npz"), **gm)

    pl = {}
    k = 0
    for (N, T, s, B) in ((1050, 200, 50, 4), (1001, 200, 50, 5), (1000, 200, 50, 256), (5000, 200, 50, 7),
                         (777, 30, 4, 10), (200, 200, 50, 4), (201, 200, 50, 4), (90, 20, 2, 3)):
        nwin = len(range(0, N - T, s))
        probs = rng.random((nwin, T, 5), dtype=np.float32)
        pl[f"par{k}"] = np.array([N, T, s, B], np.int64)
        pl[f"probs{k}"] = probs
        pl[f"merged{k}"] = ref_predict(probs, N, 5, T, s, B)
        k += 1
    pl["count"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "placement.npz"), **pl)

    # ---- 7. get_segments / yield_segments (sequence.pyx:38-53, :79-85) -----------------
    sg = {}
    k = 0
    for n in (1, 2, 3, 10, 100, 1000):
        for dens in (0.0, 0.5, 1.0):
            lab = np.zeros(n, np.int64)
            i = 0
            while i < n:
                L = int(rng.geometric(0.2))
                lab[i:i + L] = int(rng.integers(1, 5)) if rng.random() < dens else 0
                i += L
            sg[f"lab{k}"] = lab
            sg[f"all{k}"] = np.array(list(refseq.yield_segments(lab, 5)), np.int64).reshape(-1, 3)
            k += 1
    sg["count"] = np.array(k)
    np.savez_compressed(os.path.join(OUT, "segments.npz"), **sg)

    with open(os.path.join(OUT, "META.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
