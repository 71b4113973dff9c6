"""Run only the fused GRU kernel (dgrp_forward_merge) on a synthetic chromosome: for rocprofv3
counter passes and A/B timing of kernel variants."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if len(sys.argv) > 3 and sys.argv[3] == "lstm":
    rng = np.random.default_rng(0); u = int(os.environ.get("LSTM_U", "128"))
    q = np.linalg.qr(rng.normal(size=(4 * u, u)))[0].T
    m = DeviceModel(rng.uniform(-.1, .1, (5, 4 * u)), q, rng.normal(0, .05, 4 * u), rng.uniform(-.2, .2, (u, 5)), np.zeros(5), None, 200, rnn="LSTM")
    FL = 16 * u * u * 200 + 2 * u * 5 * 200
else:
    w = synthetic.trained_weights()
    if os.environ.get("GRU_ONLY_ZERO"):           # power probe: same instruction stream on all-zero operands (DVFS give-back)
        w = {k: (np.zeros_like(v) if isinstance(v, np.ndarray) else v) for k, v in w.items()}
    m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 200)
    FL = 12 * 128 * 128 * 200 + 2 * 128 * 5 * 200
st, d_idx = upload_sequence(synthetic.synthetic_chromosome(int(mbp * 1e6)))
pipe = ContigPipeline(m)
pipe.merged(d_idx); torch.cuda.synchronize()
pipe.event_log = []
for _ in range(reps):
    pipe.merged(d_idx)
torch.cuda.synchronize()
ms = [a.elapsed_time(b) for a, b, _ in pipe.event_log]
nw = pipe.event_log[0][2]
fl = nw * FL
print(f"GRU kernel: {np.mean(ms):.3f} ms (min {min(ms):.3f}) for {nw} windows = {fl/np.mean(ms)/1e9:.1f} TFLOP/s, {mbp*1e3/np.mean(ms):.0f} Mbp/s")
