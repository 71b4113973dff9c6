"""Where a FIRST command-line run of a process spends its time (a user runs `deepgrp predict` once per process): imports, library and
model load, the first launches (code-object load), first allocations, pinned slabs.  python tools/cold_start.py [Mbp]"""
import cProfile, io, os, pstats, sys, tempfile, time
t_start = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
t_torch = time.perf_counter()
from deepgrp_amd import synthetic, model as dgmodel
from deepgrp_amd.__main__ import main
t_pkg = time.perf_counter()
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 20
d = tempfile.mkdtemp()
w = synthetic.trained_weights()
mpath = os.path.join(d, "model.hdf5")
dgmodel.save_keras_hdf5(mpath, w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, vecsize=200)
fa = os.path.join(d, "a.fa")
raw = synthetic.synthetic_chromosome(int(mbp * 1e6), contig=0)
with open(fa, "wb") as fh:
    fh.write(b">chr\n" + b"\n".join(raw[i:i + 60] for i in range(0, len(raw), 60)) + b"\n")
print(f"import numpy+torch {t_torch - t_start:.2f} s, package {t_pkg - t_torch:.2f} s")
for it in range(3):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    main(["predict", mpath, fa, "--output", os.path.join(d, "out.tsv")])
    pr.disable()
    print(f"run {it}: {time.perf_counter() - t0:.3f} s")
    if it == 0:
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
        print(s.getvalue()[:6000])
