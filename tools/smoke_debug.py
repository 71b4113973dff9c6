"""where does __graft_entry__.smoke()'s row comparison part ways: stage by stage against the oracle (python tools/smoke_debug.py)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence
from oracle import oracle as orc
torch.cuda.set_device(0)
T, s = 200, 50
wts = synthetic.synthetic_weights(128, 5, False, seed=7, gain=3.0)
model = DeviceModel(wts["kernel"], wts["recurrent_kernel"], wts["bias"], wts["ff_kernel"], wts["ff_bias"], wts["scale"], vecsize=T)
seq = synthetic.synthetic_chromosome(24_000, contig=1, flank=500).decode()
st, d_idx = upload_sequence(seq.encode())
idx = d_idx.cpu().numpy()
nwin = orc.window_count(idx.size, T, s)
pipe = ContigPipeline(model, s, 256, 50, 50, True)
rows = pipe.run(seq, contig=0)
probs = model.forward_windows(d_idx, s, 0, nwin).cpu().numpy()
merged_o = orc.merge_all(probs, idx.size, s, 256)
merged_g = pipe.forward_merge(d_idx).cpu().numpy() if hasattr(pipe, "forward_merge") else None
if merged_g is not None:
    print("merged equal:", np.array_equal(merged_g.view(np.uint32), merged_o.view(np.uint32)), np.abs(merged_g - merged_o).max())
sc_o, cls_o = orc.scores(merged_o)
lab_o = orc.find_mss_labels(sc_o, cls_o.astype(np.int64), 5, 50, 50)
from deepgrp_amd._lib import check, lib
from deepgrp_amd.pipeline import stream_ptr
L = lib(); dev = d_idx.device
d_s = torch.from_numpy(sc_o).to(dev); d_c = torch.from_numpy(cls_o.astype(np.int8)).to(dev)
n = sc_o.size
lab = torch.empty(n, dtype=torch.int8, device=dev)
wb = L.dgrp_mss_workspace_bytes(n); work = torch.empty(wb, dtype=torch.uint8, device=dev)
os.environ["DGRP_MSS_TRACE"] = "1"
check(L.dgrp_mss_labels(d_s.data_ptr(), d_c.data_ptr(), n, 5, 50, 50, lab.data_ptr(), None, work.data_ptr(), wb, stream_ptr()), "mss")
torch.cuda.synchronize()
lg = lab.cpu().numpy()
print("mss labels equal on the oracle's scores:", np.array_equal(lg, lab_o), "differing positions:", np.flatnonzero(lg != lab_o)[:10], (lg != lab_o).sum())
want = orc.predict_contig(seq, lambda _i: (lambda a, b: probs[a:a + b]), T, 5, s, 256, 50, 50, True)
got = np.stack([rows["start"], rows["end"], rows["label"]], 1).reshape(-1, 3)
print("rows:", len(got), len(want))
k = 0
while k < min(len(got), len(want)) and np.array_equal(got[k], want[k]): k += 1
print("first differing row", k, got[k:k + 3].tolist(), want[k:k + 3].tolist())
