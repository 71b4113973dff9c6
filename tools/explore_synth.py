"""Explore what the synthetic model predicts (class mix, confidence, run lengths) so that the
benchmark's synthetic weights give genome-like output (mostly confident background)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
raw = synthetic.synthetic_chromosome(n, contig=0)
st, d_idx = upload_sequence(raw)
for gain, b0 in ((3.0, 0.0), (3.0, 4.0), (3.0, 6.0), (3.0, 8.0), (2.0, 6.0)):
    w = synthetic.synthetic_weights(128, 5, False, 7, gain)
    w["ff_bias"][0] += b0
    m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 200)
    pipe = ContigPipeline(m, 50, 256, 50, 50, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    merged = pipe.merged(d_idx); torch.cuda.synchronize(); t1 = time.perf_counter()
    labels = pipe.labels(merged); torch.cuda.synchronize(); t2 = time.perf_counter()
    rows = pipe.segments(labels, st); t3 = time.perf_counter()
    mx, cls = merged.max(dim=1)
    frac = torch.bincount(cls, minlength=5).float() / cls.numel()
    lab = torch.bincount(labels.long(), minlength=5).float() / cls.numel()
    print(f"gain={gain} bias0={b0}: argmax mix {np.round(frac.cpu().numpy(),3)} conf>0.99 {float((mx>0.99).float().mean()):.3f} "
          f"labels {np.round(lab.cpu().numpy(),3)} rows {len(rows)} mean_len {float((rows['end']-rows['start']).mean()) if len(rows) else 0:.1f} "
          f"| merged {1e3*(t1-t0):.1f} ms labels {1e3*(t2-t1):.1f} ms segs {1e3*(t3-t2):.1f} ms", flush=True)
    m.close()
