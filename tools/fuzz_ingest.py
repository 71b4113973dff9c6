"""Randomised FASTA texts through the device ingest against the reference's line loop (records, order, exceptions).
tools/fuzz_ingest.py [seconds] [seed]"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from deepgrp_amd.fasta import DeviceRecord, read_multi_fasta_device, read_multi_fasta_lines


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
d = tempfile.mkdtemp()
path = os.path.join(d, "f.fa")
pieces = [b"A", b"C", b"G", b"T", b"N", b"a", b"c", b"g", b"t", b"n", b"R", b"\n", b"\n", b"\n", b"\r\n", b"\r", b" ", b"\t", b">", b">h x\n", b"\n>id\n",
          b"\xc3\xa4", b"\n\n", b">\n", b"NNNN", b"ACGTACGTACGTACGTACGT" * 8]
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    k = int(rng.integers(1, 60))
    probs = np.ones(len(pieces)); probs[-1] = 8; probs[:10] = 3
    if rng.integers(0, 2):
        probs[[15, 16, 17, 21, 22]] = 0.05           # mostly clean files
    probs /= probs.sum()
    data = b"".join(pieces[i] for i in rng.choice(len(pieces), size=k, p=probs))
    if rng.integers(0, 2):
        data = b">first\n" + data
    open(path, "wb").write(data)
    want, werr = [], None
    try:
        with open(path, "r") as fh:
            for h, s in read_multi_fasta_lines(fh):
                want.append((h, s))
    except Exception as e:      # noqa: BLE001
        werr = type(e).__name__
    got, gerr = [], None
    try:
        for h, rec in read_multi_fasta_device(path, group_records=int(rng.choice([1, 2, 3, 4096]))):
            if isinstance(rec, DeviceRecord):
                got.append((h, rec.startpos, rec.length, rec.d_idx.cpu().numpy()))
            else:
                st, n = orc.strip_n(rec.encode())
                got.append((h, st, n, orc.encode_idx(rec.encode()[st:st + max(n, 0)])))
    except Exception as e:      # noqa: BLE001
        gerr = type(e).__name__
    ok = gerr == werr and len(got) == len(want)
    if ok:
        for (h, st, n, idx), (wh, ws) in zip(got, want):
            wst, wn = orc.strip_n(ws.encode())
            if (h, st, n) != (wh, wst, wn) or not np.array_equal(idx, orc.encode_idx(ws.encode()[wst:wst + max(wn, 0)])):
                ok = False
    if not ok:
        say("FAIL", repr(data), "want", werr, [(h, len(s)) for h, s in want], "got", gerr, [(g[0], g[2]) for g in got], flush=True)
        sys.exit(1)
    if it % 200 == 0:
        say(it, "files ok", flush=True)
say("done:", it, "files ok")
