"""HBM-roofline probe of the streaming kernels (encoder, one-hot windows, standalone get_max)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd._lib import check, lib
from deepgrp_amd.pipeline import stream_ptr

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

L = lib(); dev = torch.device("cuda")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
T, s, C = 200, 50, 5
seq = torch.from_numpy(np.random.default_rng(0).choice(np.frombuffer(b"ACGTN", np.uint8), size=n)).to(dev)
idx = torch.empty(n, dtype=torch.uint8, device=dev)
ms = timeit(lambda: check(L.dgrp_encode(seq.data_ptr(), n, idx.data_ptr(), stream_ptr())))
print(f"encode_kernel         {n/1e6:.0f} Mbp: {ms:.3f} ms  {2*n/ms/1e6:.0f} GB/s (2 B/bp)   {n/ms/1e3:.0f} Mbp/s")
oh = torch.empty((5, n), dtype=torch.int8, device=dev)
ms = timeit(lambda: check(L.dgrp_onehot(seq.data_ptr(), n, oh.data_ptr(), stream_ptr())))
print(f"onehot_kernel (int8)  {n/1e6:.0f} Mbp: {ms:.3f} ms  {6*n/ms/1e6:.0f} GB/s (6 B/bp)")
nwin = L.dgrp_window_count(n, T, s)
for elem, dt, name in ((2, torch.float16, "fp16"), (4, torch.float32, "fp32")):
    nw = min(nwin, (8 << 30) // (T * 5 * elem))
    out = torch.empty((nw, T, 5), dtype=dt, device=dev)
    ms = timeit(lambda: check(L.dgrp_windows_onehot(idx.data_ptr(), n, T, s, 0, nw, elem, out.data_ptr(), stream_ptr())))
    byt = nw * (T * 5 * elem + s)
    print(f"windows_kernel {name}   {nw} windows: {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s ({(T/s)*5*elem+1:.0f} B/bp)  {nw*s/ms/1e3:.0f} Mbp/s")
    del out
b = 1 << 18
probs = torch.rand((b, T, C), dtype=torch.float32, device=dev)
rows = (b - 1) * s + T
outm = torch.zeros((rows, C), dtype=torch.float32, device=dev)
ms = timeit(lambda: check(L.dgrp_get_max(outm.data_ptr(), rows, probs.data_ptr(), T, C, s, b, stream_ptr())))
byt = probs.numel() * 4 + 2 * outm.numel() * 4
print(f"get_max_kernel        {b} windows: {ms:.3f} ms  {byt/ms/1e6:.0f} GB/s ({(T/s)*C*4+2*C*4:.0f} B/bp)  {b*s/ms/1e3:.0f} Mbp/s")
