#!/usr/bin/env python3
"""Generate tests/golden/preprocessing.npz from the REFERENCE ITSELF (build container only): imports
deepgrp.preprocessing from /root/reference (pure numpy/pandas, no TensorFlow) and stores seeded inputs together
with the outputs of preprocess_y (deepgrp/preprocessing.py:9-48) and drop_start_end_n (:51-70).  Only data is
stored, nothing of the reference's source text.

Usage:  python oracle/make_golden_prep.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, "/root/reference")
import deepgrp.preprocessing as refprep                     # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    rng = np.random.default_rng(20240)
    out = {}
    # ---- preprocess_y: the whitespace separated table parse_rm.py writes (ctg start end type rep fam)
    length = 5000
    rows = []
    for _ in range(400):
        ctg = rng.choice(["chr1", "chr2", "chrX"])
        b = int(rng.integers(0, length - 10))
        e = int(min(length, b + rng.integers(1, 300)))
        typ = int(rng.integers(0, 7))
        rows.append(f"{ctg}\t{b}\t{e}\t{typ}\tRep{typ}\tFam/{typ}")
    text = "\n".join(rows) + "\n"
    out["bed_text"] = np.frombuffer(text.encode(), np.uint8)
    with tempfile.NamedTemporaryFile("w", suffix=".bed", delete=False) as fh:
        fh.write(text)
        path = fh.name
    cases = [("chr1", [1, 2, 3, 4]), ("chr2", [1, 2]), ("chrX", [2, 1, 3]), ("chr1", [3])]
    for k, (chrom, reps) in enumerate(cases):
        try:
            y = refprep.preprocess_y(path, chrom, length, list(reps))
            out[f"y{k}"] = y
            out[f"y{k}_err"] = np.array("")
        except Exception as e:      # noqa: BLE001  (a repeat number beyond the row count is an IndexError upstream)
            out[f"y{k}"] = np.zeros((0, 0), np.int8)
            out[f"y{k}_err"] = np.array(type(e).__name__)
        out[f"y{k}_chrom"] = np.array(chrom)
        out[f"y{k}_reps"] = np.array(reps)
    out["length"] = np.array(length)
    os.unlink(path)
    # ---- drop_start_end_n
    for k, (lead, body, trail) in enumerate([(7, 50, 5), (0, 40, 9), (12, 30, 0), (0, 25, 0), (3, 1, 3)]):
        n = lead + body + trail
        idx = rng.integers(0, 5, size=n)
        idx[:lead] = 4
        idx[n - trail:] = 4
        if body:
            idx[lead] = 0
            idx[lead + body - 1] = 2
        fwd = np.zeros((5, n), np.int8)
        fwd[idx, np.arange(n)] = 1
        lab = rng.integers(0, 2, size=(3, n)).astype(np.int8)
        f2, l2 = refprep.drop_start_end_n(fwd, lab)
        out[f"d{k}_fwd"], out[f"d{k}_lab"], out[f"d{k}_fwd_out"], out[f"d{k}_lab_out"] = fwd, lab, f2, l2
    np.savez_compressed(os.path.join(OUT, "preprocessing.npz"), **out)
    print("wrote", os.path.join(OUT, "preprocessing.npz"), {k: getattr(v, "shape", None) for k, v in out.items() if k.startswith("y") and not k.endswith(("err", "chrom", "reps"))})


if __name__ == "__main__":
    main()
