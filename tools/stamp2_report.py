"""Per-section cycles of gru_split2_kernel's phase from a -DDGRP_STAMP build (DGRP_STAMP_DUMP=<file> while running
tools/gru_only.py): [phase start .. barrier), the barrier, (barrier .. phase end]; per phase = per tile-step."""
import sys
import numpy as np
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 8).astype(np.float64)
a = a[a[:, 0, 3] > 0]
phases = 2 * (T - 1)
for i, nm in enumerate(["start .. barrier (67 MFMAs, 2144 pipe cycles)", "waitcnt + barrier", "barrier .. end (9 + 6 MFMAs, 384 pipe cycles)"]):
    v = a[:, :, i] / phases
    print(f"{nm:48s} mean {v.mean():7.1f}  p10 {np.percentile(v, 10):7.1f}  p90 {np.percentile(v, 90):7.1f}   by wave {v.mean(axis=0).round(1)}")
print("sum per phase", (a[:, :, :3].sum(axis=2) / phases).mean(), " loop cycles per phase", (a[:, :, 3] / phases).mean())
print("clock GHz", (a[:, :, 3] / a[:, :, 4] * 100e6).mean() / 1e9)
