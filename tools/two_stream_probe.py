"""Do the two kernels of an attention model -- the recurrent pre-pass (matrix cores / vector unit, little HBM) and the second kernel
(HBM-bound) -- overlap when the chunks of a record alternate between two streams?   python tools/two_stream_probe.py [Mbp] [units] [T]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic
from deepgrp_amd._lib import check, lib
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence, _ptr
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 20
u = int(sys.argv[2]) if len(sys.argv) > 2 else 60
T = int(sys.argv[3]) if len(sys.argv) > 3 else 342
S, B = 50, 256
L = lib()
w = synthetic.synthetic_weights(u, 5, True, seed=7, gain=1.0)
m = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"], vecsize=T)
st, d_idx = upload_sequence(synthetic.synthetic_chromosome(int(mbp * 1e6)))
n = d_idx.numel()
pipe = ContigPipeline(m, S, B, 50, 50, True)
h = pipe.handle
nwin = L.dgrp_window_count(n, T, S)
dev = d_idx.device
out = torch.zeros((n, 5), dtype=torch.float32, device=dev)


def run(nstreams, chunk):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    wb = L.dgrp_forward_workspace_bytes(h, chunk)
    works = [torch.empty(max(wb, 256), dtype=torch.uint8, device=dev) for _ in range(nstreams)]
    out.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w0, i = 0, 0
    while w0 < nwin:
        nw = min(chunk, nwin - w0)
        s_ = streams[i % nstreams]
        check(L.dgrp_forward_merge(h, _ptr(d_idx), n, S, B, w0, nw, _ptr(out), _ptr(works[i % nstreams]), works[i % nstreams].numel(),
                                   s_.cuda_stream), "fm")
        w0 += nw; i += 1
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, out.clone()


full = int(L.dgrp_forward_window_chunk(h))
os.environ["DGRP_LANE_CHUNK"] = "0"                   # the library's own lanes off: this probe makes its own
run(1, full)                                          # (warm the allocator: the first run of a size pays for its workspaces)
for chunk in (full, 32768, 16384, 8192):
    chunk = max(8192, chunk // 8192 * 8192)
    res = [min((run(k, chunk) for _ in range(3)), key=lambda x: x[0]) for k in (1, 2, 3, 4, 6)]
    same = all(bool(torch.equal(res[0][1], r[1])) for r in res[1:])
    print(f"u={u} T={T} {mbp:g} Mbp, chunks of {chunk} windows: " + ", ".join(f"{k} stream(s) {r[0]:.1f} ms ({mbp * 1e3 / r[0]:.0f} Mbp/s)"
          for k, r in zip((1, 2, 3, 4, 6), res)) + f"; merged arrays identical: {same}", flush=True)
