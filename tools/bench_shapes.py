"""Throughput of the other BASELINE model shapes (merged-probabilities stage only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence

cases = [("defaults.toml  u=60  T=342 s=50 attention", 60, 342, 50, True, 20e6),
         ("cfg5           u=256 T=500 s=25 attention", 256, 500, 25, True, 5e6),
         ("               u=256 T=500 s=25 no attention", 256, 500, 25, False, 5e6),
         ("               u=128 T=200 s=50 attention", 128, 200, 50, True, 20e6),
         ("cfg2           u=128 T=200 s=50", 128, 200, 50, False, 20e6),
         ("               u=64  T=200 s=50", 64, 200, 50, False, 20e6),
         ("               u=32  T=150 s=50 (Options defaults)", 32, 150, 50, False, 20e6),
         ("               u=96  T=200 s=50", 96, 200, 50, False, 20e6),
         ("               u=96  T=200 s=50 attention", 96, 200, 50, True, 20e6),
         ("               u=32  T=150 s=50 attention", 32, 150, 50, True, 20e6),
         # the reference's hyper-parameter space: gru_units ~ qnormal(34, 5, 2), vecsize ~ qnormal(200, 20, 2), attention
         # (/root/reference/notebooks/DeepGRP.ipynb:80,153-154)
         ("hpo            u=34  T=200 s=50 attention", 34, 200, 50, True, 20e6),
         ("hpo            u=36  T=200 s=50 attention", 36, 200, 50, True, 20e6),
         ("hpo            u=40  T=200 s=50 attention", 40, 200, 50, True, 20e6),
         ("hpo            u=44  T=200 s=50 attention", 44, 200, 50, True, 20e6),
         ("hpo            u=32  T=200 s=50 attention", 32, 200, 50, True, 20e6),
         ("               u=32  T=200 s=50", 32, 200, 50, False, 20e6),
         ("               u=34  T=200 s=50", 34, 200, 50, False, 20e6),
         ("               u=36  T=200 s=50", 36, 200, 50, False, 20e6),
         ("               u=40  T=200 s=50", 40, 200, 50, False, 20e6),
         ("               u=44  T=200 s=50", 44, 200, 50, False, 20e6),
         ("               u=48  T=200 s=50", 48, 200, 50, False, 20e6),
         ("               u=16  T=200 s=50", 16, 200, 50, False, 20e6),
         ("               u=16  T=200 s=50 attention", 16, 200, 50, True, 20e6),
         ("               u=24  T=200 s=50 attention", 24, 200, 50, True, 20e6),
         ("               u=48  T=200 s=50 attention", 48, 200, 50, True, 20e6),
         ("               u=64  T=200 s=50 attention", 64, 200, 50, True, 20e6)]
only = sys.argv[1] if len(sys.argv) > 1 else ""
only_split = os.environ.get("SHAPES_SPLIT_ONLY") == "1"
for name, u, T, s, att, n in cases:
    if only not in name:
        continue
    w = synthetic.synthetic_weights(u, 5, att, seed=7, gain=1.5)
    m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"], T)
    st, d_idx = upload_sequence(synthetic.synthetic_chromosome(int(n)))
    fl = (12 * u * u * T + 2 * (2 * u if att else u) * 5 * T + (6 * u * T if att else 0)) / s
    for fast in ((False,) if only_split else (False, True) if m.supports_split else (True,)):       # default (split operands where they exist) and --fast
        pipe = ContigPipeline(m, s, fast=fast)
        pipe.merged(d_idx); torch.cuda.synchronize()
        t0 = time.perf_counter(); pipe.merged(d_idx); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        kind = "split" if pipe.split else "fp16 "
        print(f"{name:52s} {kind} {n/1e6:4.0f} Mbp: {dt*1e3:8.1f} ms  {n/dt/1e6:7.1f} Mbp/s  {fl*n/dt/1e12:6.1f} TFLOP/s", flush=True)
    m.close(); del d_idx, pipe; torch.cuda.empty_cache()
