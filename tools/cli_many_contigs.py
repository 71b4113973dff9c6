"""Command line on a draft-assembly-like FASTA: many short records (file -> TSV file, warm)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic, model as dgmodel
from deepgrp_amd.__main__ import main

ncontig = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
kbp = float(sys.argv[2]) if len(sys.argv) > 2 else 10
d = tempfile.mkdtemp()
w = synthetic.trained_weights()
mpath = os.path.join(d, "model.hdf5")
dgmodel.save_keras_hdf5(mpath, w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, vecsize=200)
fa = os.path.join(d, "asm.fa")
raw = synthetic.synthetic_chromosome(int(ncontig * kbp * 1e3) + 40000, contig=0)[20000:-20000]   # without the N blocks at both ends
n = int(kbp * 1e3)
with open(fa, "wb") as fh:
    for k in range(ncontig):
        seq = raw[k * n:(k + 1) * n]
        fh.write(b">ctg%d\n" % (k + 1))
        fh.write(b"\n".join(seq[i:i + 60] for i in range(0, len(seq), 60)) + b"\n")
mbp = ncontig * kbp / 1e3
for it in range(2):
    t0 = time.perf_counter()
    main(["predict", mpath, fa, "--output", os.path.join(d, "out.tsv")])
    dt = time.perf_counter() - t0
    print(f"run {it}: {ncontig} records x {kbp:g} kbp = {mbp:g} Mbp -> TSV in {dt:.3f} s = {mbp/dt:.1f} Mbp/s, {sum(1 for _ in open(os.path.join(d,'out.tsv')))} rows", flush=True)
