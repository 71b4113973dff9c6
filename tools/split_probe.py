"""Split-operand fused kernel against the fp32 kernels and (on a few windows) the float64 CPU statement."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import DeviceModel, upload_sequence
from oracle import oracle as orc
_st, d_idx = upload_sequence(synthetic.synthetic_chromosome(200 + 50 * 2048 + 1000, contig=0, flank=500))
idx = d_idx.cpu().numpy()
for name, w in (("trained", synthetic.trained_weights()), ("gain3", synthetic.synthetic_weights(128, 5, False, 7, 3.0)),
                ("u60", synthetic.synthetic_weights(60, 5, False, 3, 2.0)), ("u20", synthetic.synthetic_weights(20, 5, False, 3, 2.0)),
                ("att128g2", synthetic.synthetic_weights(128, 5, True, 9, 2.0)), ("att128g3", synthetic.synthetic_weights(128, 5, True, 5, 3.0)),
                ("att60", synthetic.synthetic_weights(60, 5, True, 3, 1.0))):
    dm = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"], vecsize=200)
    ow = orc.Weights(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"], 200)
    o = orc.nn_forward(idx, ow, 50, 0, 64, np.float64)
    for level in (0, 1):
        dm.set_precision(level)
        r = dm.check_accuracy(d_idx, 50, 2048)
        f = dm.forward_windows(d_idx, 50, 0, 64).cpu().numpy()
        print(name, "flags", dm.kernel_flags, "level", level, "vs fp32 kernels: max %.3e median %.2e q99 %.2e above1e-3 %d | vs float64 (64 windows): %.3e"
              % (r["max_abs_diff"], r["median_window_max"], r["q99_window_max"], r["positions_above_1e-3"], np.abs(f - o).max()), flush=True)
    dm.close()
