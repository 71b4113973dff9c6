"""Two forward modes against each other on a whole synthetic chromosome with the benchmark's model: distribution of the
per-base probability deviation, label agreement, TSV rows, and the speed of both.
python tools/precise_vs_fast.py [Mbp] [trained|random] [fp16-vs-split | split-vs-fp32 | fp16-vs-fp32]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 10
kind = sys.argv[2] if len(sys.argv) > 2 else "trained"
w = synthetic.trained_weights() if kind == "trained" else synthetic.synthetic_weights(128, 5, False, 7, 3.0)
m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"], vecsize=200)
st, d_idx = upload_sequence(synthetic.synthetic_chromosome(int(mbp * 1e6)))
what = sys.argv[3] if len(sys.argv) > 3 else "fp16-vs-split"          # or split-vs-fp32, fp16-vs-fp32
def make(kind):
    if kind == "fp16":
        return ContigPipeline(m, fast=True)
    return ContigPipeline(m, fp32=(kind == "fp32"))     # "split" = the default; "fp32" = the plain-fp32 kernels
a_kind, b_kind = what.split("-vs-")
fast, precise = make(a_kind), make(b_kind)
print(f"A = {a_kind} ({'fused, fp16 operands' if a_kind == 'fp16' else 'fused, split operands'}), "
      f"B = {b_kind} ({'plain-fp32 kernels' if precise.fp32 else 'fused, split operands'})", flush=True)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
fast.merged(d_idx)
t0 = T(); mf = fast.merged(d_idx); t1 = T(); precise.merged(d_idx); t1b = T(); mp = precise.merged(d_idx); t2 = T(); t1 = t1 - (t1b - t1) * 0; t0p = t1b
print(f"{mbp:g} Mbp {kind}: A forward+merge {1e3*(t1-t0):.1f} ms ({mbp/(t1-t0):.0f} Mbp/s); B {1e3*(t2-t0p):.1f} ms ({mbp/(t2-t0p):.1f} Mbp/s)", flush=True)
d = (mf - mp).abs().amax(dim=1)
q = torch.quantile(d[:: max(1, d.numel() // 4_000_000)].double(), torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9999], dtype=torch.float64, device=d.device)).cpu().numpy()
print("per-base max|dp| of the merged probabilities: median %.2e  q90 %.2e  q99 %.2e  q99.9 %.2e  q99.99 %.2e  worst %.2e;  bases above 1e-3: %d of %d (%.4f %%)"
      % (*q, float(d.max()), int((d > 1e-3).sum()), d.numel(), 100.0 * float((d > 1e-3).sum()) / d.numel()), flush=True)
print("argmax differs on %d bases" % int((mf.argmax(dim=1) != mp.argmax(dim=1)).sum()))
lf, lp = fast.labels(mf), precise.labels(mp)
print("final labels differ on %d bases" % int((lf != lp).sum()))
rf, rp = fast.segments(lf, st), precise.segments(lp, st)
same = len(rf) == len(rp) and bool((rf == rp).all())
print("TSV rows: A %d, B %d, identical: %s" % (len(rf), len(rp), same))
if not same:
    sf = {(int(a), int(b), int(c)) for a, b, c in zip(rf["start"], rf["end"], rf["label"])}
    sp = {(int(a), int(b), int(c)) for a, b, c in zip(rp["start"], rp["end"], rp["label"])}
    print("rows only in A: %d, only in B: %d; examples %s | %s" % (len(sf - sp), len(sp - sf), sorted(sf - sp)[:3], sorted(sp - sf)[:3]))
