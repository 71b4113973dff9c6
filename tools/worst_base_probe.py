"""Locate the bases where the fast and the precise forward differ most and check both against the float64 CPU checker."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence
from oracle import oracle as orc

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50
w = synthetic.trained_weights()
m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"], vecsize=200)
ow = orc.Weights(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 200)
st, d_idx = upload_sequence(synthetic.synthetic_chromosome(int(mbp * 1e6)))
idx = d_idx.cpu().numpy()
mf = ContigPipeline(m, fast=True).merged(d_idx); mp = ContigPipeline(m, precise=True).merged(d_idx)
d = (mf - mp).abs().amax(dim=1)
top = torch.topk(d, 8).indices.cpu().numpy()
for pos in sorted(top):
    wlo = max(0, (pos - 199 + 49) // 50); whi = min(pos // 50, orc.window_count(idx.size, 200, 50) - 1)
    nw = whi - wlo + 1
    m.set_precision(0)
    f = m.forward_windows(d_idx, 50, wlo, nw).cpu().numpy()                 # the fp16-operand kernel
    p = m.forward_windows_reference(d_idx, 50, wlo, nw).cpu().numpy()
    o = orc.nn_forward(idx, ow, 50, wlo, nw, np.float64)
    print("base %d dp %.3e  context %s" % (pos, float(d[pos]), "".join("ACGTN"[c] for c in idx[max(0, pos - 12):pos + 12])))
    for k in range(nw):
        t = pos - (wlo + k) * 50
        if 0 <= t < 200:
            print("   window %d t=%3d  fast %s  precise %s  f64 %s   |fast-f64| %.2e |precise-f64| %.2e   window-wide max: fast %.2e precise %.2e"
                  % (wlo + k, t, np.round(f[k, t], 4), np.round(p[k, t], 4), np.round(o[k, t], 4), np.abs(f[k, t] - o[k, t]).max(),
                     np.abs(p[k, t] - o[k, t]).max(), np.abs(f[k] - o[k]).max(), np.abs(p[k] - o[k]).max()))
