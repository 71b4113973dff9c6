"""Randomised score arrays through dgrp_mss_labels (single record, multi-stretch machinery) and dgrp_mss_labels_batch against
the oracle's mss_find_all + vote.  tools/fuzz_mss.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from deepgrp_amd._lib import check, lib
from deepgrp_amd.pipeline import require_gpu, stream_ptr


def say(*a, **_kw):
    """Progress goes to stdout AND to gpurun_out/<tool>.progress: a long sweep behind a pipe (`| tail`) shows no output until
    the pipe ends, which gpurun takes for a hang (profiles/r01_fuzz_summary.txt's run was killed that way)."""
    import os as _os
    line = " ".join(str(x) for x in a)
    print(line, flush=True)
    _os.makedirs("gpurun_out", exist_ok=True)
    with open(_os.path.join("gpurun_out", _os.path.basename(__file__)[:-3] + ".progress"), "a") as fh:
        fh.write(line + "\n")



budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev, L = require_gpu(), lib()


def scores(n):
    style = int(rng.integers(0, 6))
    if style == 0:                                # confident runs (the real thing)
        lab = np.resize(np.repeat(rng.integers(0, 5, size=n // 30 + 2), rng.integers(1, 400, size=n // 30 + 2)), n)
        m = np.clip(rng.uniform(0.3, 0.9999, n), None, 0.99).astype(np.float32)
        t = np.abs(np.log(m / (1 - m)))
        return np.where(lab > 0, t, -10 * t).astype(np.float64), lab.astype(np.int64)
    if style == 1:
        return rng.normal(float(rng.normal(0, 1)), 5, n), rng.integers(0, 5, n).astype(np.int64)
    if style == 2:
        return rng.integers(-6, 7, n).astype(np.float64) * 0.5, rng.integers(0, 3, n).astype(np.int64)
    if style == 3:                                # long background with rare hits
        s = -np.abs(rng.normal(40, 10, n))
        hits = rng.integers(0, n, size=max(1, n // 5000))
        for h in hits:
            s[h:h + int(rng.integers(1, 300))] = rng.uniform(1, 5)
        return s, rng.integers(0, 5, n).astype(np.int64)
    if style == 4:                                # huge magnitudes next to tiny ones: exactness of the certificate
        s = rng.normal(0, 1, n) * (10.0 ** rng.integers(-8, 9, n))
        return s, rng.integers(0, 5, n).astype(np.int64)
    s = np.where(rng.random(n) < 0.5, 4.59, -45.95) + rng.normal(0, 1e-6, n)      # saturated scores, drifting sums
    return s, rng.integers(0, 5, n).astype(np.int64)


t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    ml, xd = int(rng.choice([0, 1, 3, 10, 50])), int(rng.choice([-1, 0, 1, 10, 50]))
    sub = int(rng.choice([0, 0, 1, 2, 7, 64, 256, 500]))  # blocks per light unit (0: the built-in 2 048): state handed from unit to unit
    os.environ.pop("DGRP_MSS_SUB", None)
    if sub:
        os.environ["DGRP_MSS_SUB"] = str(sub)
    if rng.integers(0, 3) == 0:                   # one big record through the stretch machinery
        n = int(rng.choice([1, 63, 64, 65, 4096, 100_000, 1_000_000, 2_000_003]))
        S, lab = scores(n)
        want = orc.find_mss_labels(S, lab, 5, ml, xd)
        d_S = torch.from_numpy(S).to(dev); d_c = torch.from_numpy(lab.astype(np.int8)).to(dev)
        out = torch.empty(n, dtype=torch.int8, device=dev)
        wb = L.dgrp_mss_workspace_bytes(n); work = torch.empty(wb, dtype=torch.uint8, device=dev)
        check(L.dgrp_mss_labels(d_S.data_ptr(), d_c.data_ptr(), n, 5, ml, xd, out.data_ptr(), None, work.data_ptr(), wb, stream_ptr()), "mss")
        ok = np.array_equal(out.cpu().numpy(), want)
        what = f"single n={n}"
    else:
        lens = [int(x) for x in rng.choice([1, 2, 63, 64, 65, 1000, 5000, 70_000, 250_000], size=int(rng.integers(1, 12)))]
        starts = np.zeros(len(lens) + 1, np.int64)
        for i, n in enumerate(lens):
            starts[i + 1] = starts[i] + (n + 63) // 64 * 64
        total = int(starts[-1])
        S = np.zeros(total); cls = np.zeros(total, np.int8); want = np.zeros(total, np.int8)
        for i, n in enumerate(lens):
            s, lab = scores(n); a = int(starts[i])
            S[a:a + n] = s; cls[a:a + n] = lab; want[a:a + n] = orc.find_mss_labels(s, lab, 5, ml, xd)
        d_S = torch.from_numpy(S).to(dev); d_c = torch.from_numpy(cls).to(dev)
        out = torch.empty(total, dtype=torch.int8, device=dev)
        wb = L.dgrp_mss_batch_workspace_bytes(total, len(lens)); work = torch.empty(wb, dtype=torch.uint8, device=dev)
        check(L.dgrp_mss_labels_batch(d_S.data_ptr(), d_c.data_ptr(), total, len(lens), starts.ctypes.data, 5, ml, xd, out.data_ptr(),
                                      work.data_ptr(), wb, stream_ptr()), "mss batch")
        got = out.cpu().numpy()
        ok = all(np.array_equal(got[int(starts[i]):int(starts[i]) + n], want[int(starts[i]):int(starts[i]) + n]) for i, n in enumerate(lens))
        what = f"batch lens={lens}"
    if not ok:
        say("FAIL", what, "ml/xd", ml, xd, "sub", sub, flush=True)
        sys.exit(1)
    if it % 50 == 0:
        say(it, "cases ok", flush=True)
say("done:", it, "cases ok")
