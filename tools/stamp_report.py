"""Diagnostic: per-segment cycle stamps of the fused GRU kernel (build with GRU_EXTRA=-DDGRP_STAMP,
run with DGRP_STAMP_DUMP=<file>), averaged per step."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 16).astype(np.float64)
T = 200
names = ["barrier->top", "xa,loads,r chain,dense issue", "finish_step(t-2)", "g chain+sig(r)", "z chain+r*g+tanh", "sig(z)+blend+publish", "barrier wait"]
full = a[(a[:, 0, 8] > 0)]
print("workgroups", len(full))
seg = full[:, :, :7] / T
for k, nm in enumerate(names):
    print(f"{nm:22s} mean {seg[:, :, k].mean():8.1f}  p10 {np.percentile(seg[:, :, k], 10):8.1f}  p90 {np.percentile(seg[:, :, k], 90):8.1f}")
print("sum per step", seg.sum(axis=2).mean(), " loop cycles/T", (full[:, :, 8] / T).mean())
clk = full[:, :, 8] / full[:, :, 9] * 100e6
print("clock GHz", clk.mean() / 1e9, clk.min() / 1e9, clk.max() / 1e9)
print("prologue cycles", full[:, :, 10].mean(), " loop", full[:, :, 8].mean(), " whole kernel", full[:, :, 11].mean(),
      " epilogue", (full[:, :, 11] - full[:, :, 10] - full[:, :, 8]).mean())

st = full[:, 0, 12]; en = full[:, 0, 13]
t0 = st.min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0     # microseconds
print("kernel span us", en.max(), " mean WG lifetime us", (en - st).mean(), " sum lifetimes / span = mean concurrency", (en - st).sum() / en.max())
hw = full[:, 0, 14].astype(np.int64)
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; waveid = hw & 0xf; simd = (hw >> 4) & 3
print("wave_id hist", np.bincount(waveid), " simd hist", np.bincount(simd))
# gaps on a slot: group by (blockIdx order unknown) -> use (se, sh, cu, waveid) as slot key (+xcc unknown)
import collections
key = hw & 0xffff
order = np.argsort(st)
last = {}
gaps = []
for i in order:
    k = int(key[i])
    if k in last: gaps.append(st[i] - last[k])
    last[k] = en[i]
gaps = np.array(gaps)
print("slots", len(last), " gap between consecutive WGs on a slot key us: mean", gaps.mean(), " p50", np.median(gaps), " p90", np.percentile(gaps, 90))
