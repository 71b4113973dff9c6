"""End-to-end wall clock of the command line on a synthetic FASTA file (file -> TSV file)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic, model as dgmodel
from deepgrp_amd.__main__ import main

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50
d = tempfile.mkdtemp()
w = synthetic.trained_weights()
mpath = os.path.join(d, "model.hdf5")
dgmodel.save_keras_hdf5(mpath, w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, vecsize=200)
fa = os.path.join(d, "chr.fa")
with open(fa, "wb") as fh:
    for k in range(2):
        raw = synthetic.synthetic_chromosome(int(mbp * 1e6 / 2), contig=k)
        fh.write(b">chr%d\n" % (k + 1))
        fh.write(b"\n".join(raw[i:i + 60] for i in range(0, len(raw), 60)) + b"\n")
for it in range(2):
    t0 = time.perf_counter()
    main(["predict", mpath, fa, "--output", os.path.join(d, "out.tsv")])
    dt = time.perf_counter() - t0
    print(f"run {it}: {mbp:g} Mbp FASTA -> TSV in {dt:.3f} s = {mbp/dt:.0f} Mbp/s, {sum(1 for _ in open(os.path.join(d,'out.tsv')))} rows", flush=True)
