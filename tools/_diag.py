import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import DeviceModel, upload_sequence
w = synthetic.synthetic_weights(128, 5, False, seed=7, gain=1.0)
m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 200)
rng = np.random.default_rng(0)
seq = bytes(rng.choice(list(b"ACGT"), size=200 + 50 * 69))
st, d_idx = upload_sequence(seq)
for nw in (64, 70, 32):
    got = m.forward_windows(d_idx, 50, 0, nw).cpu().numpy()
    ref = m.forward_windows_reference(d_idx, 50, 0, nw).cpu().numpy()
    nan = np.isnan(got).any(axis=2)            # [nw, T]
    print("nw", nw, "windows with nan:", np.nonzero(nan.any(axis=1))[0].tolist())
    for wi in np.nonzero(nan.any(axis=1))[0][:6]:
        print("  window", wi, "nan steps:", np.nonzero(nan[wi])[0][:20].tolist(), "count", int(nan[wi].sum()))
    ok = ~nan
    d = np.abs(got - ref).max(axis=2)
    print("  max diff over non-nan:", float(d[ok].max()) if ok.any() else None)
    bad = (d > 1e-4) & ok
    print("  non-nan positions off by >1e-4:", int(bad.sum()), [ (int(a), int(b)) for a, b in zip(*np.nonzero(bad)) ][:10])
