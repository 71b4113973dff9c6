"""one stitched-pieces case outside pytest (device printf of a -DMSS_DEBUG build shows up): python tools/mss_debug_case.py drift exact xd"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DGRP_MSS_SUB"] = "256"; os.environ["DGRP_MSS_TRACE"] = "1"
from deepgrp_amd._lib import check, lib
from deepgrp_amd.pipeline import require_gpu, stream_ptr
from oracle import oracle as orc
drift, exact, xd = float(sys.argv[1]), sys.argv[2] == "1", int(sys.argv[3])
dev, L = require_gpu(), lib()
rng = np.random.default_rng(int(drift * 100) + xd + (17 if exact else 0))
n = 300_007
scores = rng.normal(drift, 1.0, size=n)
if exact:
    scores = np.round(scores * 1024) / 1024
else:
    scores[::7] *= 2.0 ** -17
    scores[:2000] += 3000.0
scores[rng.random(n) < 0.01] = 0.0
scores[140_000:140_200] = -40.0
scores[200_000:200_040] = -3.0
cls = rng.integers(0, 5, size=n).astype(np.int64)
want, segs = orc.find_mss_labels(scores, cls, 5, 3, xd, return_segments=True)
d_s = torch.from_numpy(scores).to(dev); d_l = torch.from_numpy(cls.astype(np.int8)).to(dev)
lab = torch.empty(n, dtype=torch.int8, device=dev)
wb = L.dgrp_mss_workspace_bytes(n)
work = torch.empty(wb, dtype=torch.uint8, device=dev)
nseg = torch.zeros(1, dtype=torch.int64, device=dev)
check(L.dgrp_mss_labels(d_s.data_ptr(), d_l.data_ptr(), n, 5, 3, xd, lab.data_ptr(), nseg.data_ptr(), work.data_ptr(), wb, stream_ptr()), "mss")
torch.cuda.synchronize()
print("segments", int(nseg.item()), "oracle", len(segs), "labels equal", bool(np.array_equal(lab.cpu().numpy(), want)))
