"""Randomised command-line runs (random model, flags, multi-record FASTA) against what the reference pipeline would print,
computed by the oracle from the GPU's own probabilities.  tools/fuzz_cli.py [seconds] [seed]"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from deepgrp_amd import model as dgmodel
from deepgrp_amd.__main__ import main
from deepgrp_amd.pipeline import upload_sequence


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
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    it += 1
    u = int(rng.choice([8, 16, 33, 64, 128]))
    T = int(rng.choice([5, 20, 30, 64, 100]))
    att = bool(rng.integers(0, 2))
    lstm = (not att) and bool(rng.integers(0, 4) == 0)
    mpath = os.path.join(d, "m.hdf5")
    if lstm:
        w = orc.LSTMWeights.random(u, 5, T, seed=int(rng.integers(0, 1 << 30)), gain=1.5)
        dgmodel.save_keras_hdf5(mpath, w.kernel, w.recurrent, w.bias, w.ff_kernel, w.ff_bias, None, vecsize=T, rnn="LSTM")
    else:
        w = orc.Weights.random(u, 5, T, att, seed=int(rng.integers(0, 1 << 30)), gain=float(rng.choice([1.0, 2.0, 3.0])))
        dgmodel.save_keras_hdf5(mpath, w.kernel, w.recurrent, w.bias, w.ff_kernel, w.ff_bias, w.scale, vecsize=T)
    s, B = int(rng.integers(1, T + 5)), int(rng.choice([1, 3, 7, 256]))
    ml, xd = int(rng.choice([0, 1, 3, 50])), int(rng.choice([-1, 0, 1, 5, 50]))
    use_mss = bool(rng.integers(0, 5) != 0)
    nrec = int(rng.integers(1, 40))
    recs = []
    for k in range(nrec):
        n = int(rng.choice([1, 2, T - 1 if T > 1 else 1, T, T + 1, 64, 200, 1500, 7000]))
        seq = "".join(rng.choice(list("ACGTNacgtn"), size=n, p=[.2, .2, .2, .2, .02, .04, .04, .04, .04, .02]))
        if not seq.strip("Nn"):
            seq = "A" + seq                                       # all-N records raise (covered by the tests)
        wrap = int(rng.choice([50, 60, 70, 10_000]))
        nl = "\r\n" if rng.integers(0, 6) == 0 else "\n"
        recs.append(f">rec{k} d={n}{nl}" + nl.join(seq[i:i + wrap] for i in range(0, len(seq), wrap)) + nl)
    if rng.integers(0, 5) == 0:
        recs.insert(int(rng.integers(0, len(recs) + 1)), ">odd\nAC GT\nacgtacgtacgtacgtacgtacgtacgt\n")
    fa = os.path.join(d, "f.fa")
    open(fa, "w", newline="").write("".join(recs))
    out = os.path.join(d, "o.tsv")
    argv = ["-b", str(B), "-s", str(s), "-x", str(xd), "-l", str(ml), "predict", mpath, fa, "--output", out] + ([] if use_mss else ["-m"])
    tag = f"it {it}: u={u} T={T} att={att} lstm={lstm} argv={argv[:8]} use_mss={use_mss} nrec={len(recs)}"
    try:
        main(argv)
        model = dgmodel.load_model(mpath)
        want = []
        with open(fa) as fh:
            for header, seq in orc.read_multi_fasta(fh):
                st, d_idx = upload_sequence(seq.encode())
                nwin = orc.window_count(d_idx.numel(), T, s)
                probs = model.forward_windows(d_idx, s, 0, nwin).cpu().numpy() if nwin else np.zeros((0, T, 5), np.float32)
                rows = orc.predict_contig(seq, lambda _i: (lambda a, b: probs[a:a + b]), T, 5, s, B, ml, xd, use_mss)
                want += [f"{fa}\t{header}\t{a}\t{b}\t{c}\n" for a, b, c in rows]
        model.close()
        got = open(out).read()
        assert got == "".join(want), "TSV differs"
    except Exception as e:      # noqa: BLE001
        say("FAIL", tag, "->", repr(e)[:300], flush=True)
        import shutil; shutil.copy(fa, "gpurun_out/fuzz_cli_fail.fa")
        sys.exit(1)
    if it % 20 == 0:
        say(it, "runs ok", flush=True)
say("done:", it, "runs ok")
