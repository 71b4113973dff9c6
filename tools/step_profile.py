import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from deepgrp_amd import synthetic
from deepgrp_amd._lib import check, lib
from deepgrp_amd.distributed import gather_records
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence, stream_ptr
w = synthetic.trained_weights()
m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, 200)
raw = synthetic.synthetic_chromosome(50_000_000)
st, d_idx = upload_sequence(raw)
d_seq = torch.from_numpy(np.frombuffer(raw, np.uint8)[st:st + d_idx.numel()].copy()).cuda()
pipe = ContigPipeline(m); n = d_seq.numel(); dev = torch.device("cuda")
def step():
    check(lib().dgrp_encode(d_seq.data_ptr(), n, d_idx.data_ptr(), stream_ptr()))
    rows = pipe.run_idx(d_idx, st, contig=0)
    return gather_records(rows, dev)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print(f"step {i}: {1e3*(time.perf_counter()-t0):.2f} ms", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
