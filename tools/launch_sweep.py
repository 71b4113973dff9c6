"""the fused forward + merge of the benchmark model at different launch sizes (windows per launch: the "internal GPU
batch"), one 250 Mbp record, the whole record per measurement.  python tools/launch_sweep.py [Mbp] [sizes...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 250
sizes = [int(x) for x in sys.argv[2:]] or [1024, 4096, 16384, 65536, 262144, 1 << 20]
w = synthetic.trained_weights()
m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 200)
st, d_idx = upload_sequence(synthetic.synthetic_chromosome(int(mbp * 1e6)))
FL = 12 * 128 * 128 * 200 + 2 * 128 * 5 * 200
ref = None
for chunk in sizes:
    pipe = ContigPipeline(m)
    pipe.chunk_windows = chunk
    out = pipe.merged(d_idx)
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    same = bool(torch.equal(out, ref))
    del out
    t0 = time.perf_counter()
    pipe.merged(d_idx)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nwin = len(range(0, d_idx.numel() - 200, 50))
    print(f"{chunk:8d} windows per launch ({(chunk + 31) // 32:6d} workgroups, {-(-nwin // chunk):5d} launches): {dt * 1e3:8.1f} ms  "
          f"{mbp / dt:6.1f} Mbp/s  {nwin * FL / dt / 1e12:6.1f} TFLOP/s  merged rows {'identical' if same else 'DIFFER'}", flush=True)
