"""dgrp_mss_labels on score arrays WITHOUT the structure of a genome (the serial cliff of VERDICT r01 #5): 10 M scores per style,
wall clock of the whole labels stage and -- with DGRP_MSS_TRACE=1 -- stretches / light units / passes.
    python tools/mss_cliff.py [Mscores] [check|nocheck] [style substring]      (check: compare with the oracle; minutes of CPU for 10 M)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd._lib import check, lib
from deepgrp_amd.pipeline import require_gpu, stream_ptr

n = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 10_000_000
verify = len(sys.argv) > 2 and sys.argv[2] == "check"
dev, L = require_gpu(), lib()
rng = np.random.default_rng(5)


def dirichlet_scores(alpha):
    """scores of Dirichlet class probabilities, as deepgrp/prediction.py:51-59 computes them"""
    p = rng.dirichlet(np.full(5, alpha), size=n).astype(np.float32)
    m = np.clip(p.max(axis=1), None, 0.99)
    t = np.abs(np.log(m / (1 - m))).astype(np.float64)
    return np.where(p.argmax(axis=1) > 0, t, -10 * t), p.argmax(axis=1)


styles = {
    "dirichlet(0.3) probabilities: downward drift (class 0 scores -10 t), L ~ -1e7 next to 2^-40 quanta": lambda: dirichlet_scores(0.3),
    "normal(+0.3, 1): upward drift, inexact": lambda: (rng.normal(0.3, 1, n), rng.integers(0, 5, n)),
    "normal(0, 5): no drift": lambda: (rng.normal(0, 5, n), rng.integers(0, 5, n)),
    "normal(-0.3, 1): downward drift (a flush at almost every run)": lambda: (rng.normal(-0.3, 1, n), rng.integers(0, 5, n)),
    "normal(-0.02, 1): slow downward drift, x-drop resets": lambda: (rng.normal(-0.02, 1, n), rng.integers(0, 5, n)),
}
only = sys.argv[3] if len(sys.argv) > 3 else ""
for name, make in styles.items():
    if only not in name:
        continue
    S, cls = make()
    d_S = torch.from_numpy(np.ascontiguousarray(S, np.float64)).to(dev)
    d_c = torch.from_numpy(cls.astype(np.int8)).to(dev)
    out = torch.empty(n, dtype=torch.int8, device=dev)
    nseg = torch.zeros(1, dtype=torch.int64, device=dev)
    wb = L.dgrp_mss_workspace_bytes(n)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(L.dgrp_mss_labels(d_S.data_ptr(), d_c.data_ptr(), n, 5, 50, 50, out.data_ptr(), nseg.data_ptr(), work.data_ptr(), wb,
                                stream_ptr()), "mss")
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
        if rep == 0:
            os.environ.pop("DGRP_MSS_TRACE", None)
    os.environ["DGRP_MSS_TRACE"] = "1" if os.environ.get("MSS_CLIFF_TRACE") else ""
    if not os.environ["DGRP_MSS_TRACE"]:
        os.environ.pop("DGRP_MSS_TRACE")
    line = f"{name}: {best * 1e3:.1f} ms, {int(nseg.item())} segments"
    if verify:
        from oracle import oracle as orc
        want = orc.find_mss_labels(np.ascontiguousarray(S, np.float64), cls.astype(np.int64), 5, 50, 50)
        line += "  == oracle" if np.array_equal(out.cpu().numpy(), want) else "  DIFFERS FROM THE ORACLE"
    print(line, flush=True)
