"""Per-stage wall clock of one record through the device pipeline (sync after each stage)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic
from deepgrp_amd._lib import check, lib
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence, stream_ptr

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50
w = synthetic.trained_weights() if (len(sys.argv) < 3 or sys.argv[2] == "trained") else synthetic.synthetic_weights(128, 5, False, 7, 3.0)
m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 200)
raw = synthetic.synthetic_chromosome(int(mbp * 1e6))
st, d_idx = upload_sequence(raw)
d_seq = torch.from_numpy(np.frombuffer(raw, np.uint8)[st:st + d_idx.numel()].copy()).cuda()
pipe = ContigPipeline(m)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    t0 = T()
    check(lib().dgrp_encode(d_seq.data_ptr(), d_seq.numel(), d_idx.data_ptr(), stream_ptr()))
    t1 = T(); merged = pipe.merged(d_idx)
    t2 = T(); labels = pipe.labels(merged)
    t3 = T(); rows = pipe.segments(labels, st)
    t4 = T()
    print(f"iter {it}: encode {1e3*(t1-t0):.2f} merged {1e3*(t2-t1):.2f} labels {1e3*(t3-t2):.2f} segments {1e3*(t4-t3):.2f} total {1e3*(t4-t0):.2f} ms rows {len(rows)}", flush=True)
