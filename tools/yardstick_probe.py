import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import DeviceModel, upload_sequence
_st, d_idx = upload_sequence(synthetic.synthetic_chromosome(200 + 50 * 2048 + 1000, contig=0, flank=500))
for name, w in (("trained", synthetic.trained_weights()), ("gain3", synthetic.synthetic_weights(128, 5, attention=False, seed=7, gain=3.0)),
          ("gain2att", synthetic.synthetic_weights(128, 5, attention=True, seed=9, gain=2.0)), ("gain1", synthetic.synthetic_weights(128, 5, attention=False, seed=7, gain=1.0))):
    dm = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"], vecsize=200)
    diffs = []
    for w0 in range(0, 2048, 256):
        fast = dm.forward_windows(d_idx, 50, w0, 256)
        ref = dm.forward_windows_reference(d_idx, 50, w0, 256)
        diffs.append((fast - ref).abs().amax(dim=2))     # [256, T]
    d = torch.cat(diffs).cpu().numpy()
    per_win = d.max(axis=1)
    print(name, "flags", dm.kernel_flags, "worst %.3e" % d.max(), "windows >1e-3:", int((per_win > 1e-3).sum()), "of", len(per_win),
          "quantiles of per-window max", np.quantile(per_win, [0.5, 0.9, 0.99, 0.999]).round(6), "argmax t of worst", int(d.max(axis=0).argmax()))
    dm.close()
