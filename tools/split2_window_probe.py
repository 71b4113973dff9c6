"""gru_split2_kernel (window probabilities, MODE 1) against the float64 oracle, error by window and by step: which tile of a
workgroup and which steps a wrong schedule variant breaks (tools/build_variant.sh, tools/lint_split2_isa.py).

    python tools/split2_window_probe.py [out.npy]
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from deepgrp_amd.pipeline import DeviceModel
u, T, s, nw = 128, 200, 50, 70
rng = np.random.default_rng(u * 1000 + T)
w = orc.Weights.random(u, 5, T, False, seed=7, gain=1.0)
dm = DeviceModel(w.kernel, w.recurrent, w.bias, w.ff_kernel, w.ff_bias, w.scale, vecsize=T)
n = (nw + 2) * s + T
idx = rng.choice(5, size=n, p=[0.24, 0.25, 0.25, 0.24, 0.02]).astype(np.uint8)
want = orc.nn_forward(idx, w, s, 2, nw, np.float64)
dm.set_precision(1)
got = dm.forward_windows(torch.from_numpy(idx).cuda(), s, 2, nw).cpu().numpy()
err = np.abs(got - want).max(axis=2)       # [nw, T]
print("max err", err.max())
bw, bt = np.nonzero(err > 1e-5)
print("bad windows:", sorted(set(bw.tolist())))
print("bad steps:", sorted(set(bt.tolist()))[:50])
for wv in sorted(set(bw.tolist()))[:6]:
    print(wv, np.nonzero(err[wv] > 1e-5)[0][:20], err[wv].max())
if len(sys.argv) > 1:
    np.save(sys.argv[1], got)
