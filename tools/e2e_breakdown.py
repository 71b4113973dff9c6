"""Where the end-to-end leg of bench.py spends its time: FASTA file in /dev/shm -> TSV bytes, one record of [Mbp] (default 250).
Prints wall clock per repetition, then a cProfile table (cumulative) of one more repetition.
    python tools/e2e_breakdown.py [Mbp]"""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd.fasta import read_multi_fasta_device
from deepgrp_amd import synthetic
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel
from deepgrp_amd.runner import RecordRunner, rows_text, rows_text_batch

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 250
n = int(mbp * 1e6)
raw = synthetic.synthetic_chromosome(n, contig=0)
w = synthetic.trained_weights()
model = DeviceModel(w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"], vecsize=200)
pipe = ContigPipeline(model, 50, 256, 50, 50, use_mss=True)
path = f"/dev/shm/dgrp_e2e_{os.getpid()}.fa"
body = np.frombuffer(raw, np.uint8)
full = body.size // 60 * 60
lines = np.empty((full // 60, 61), np.uint8)
lines[:, :60] = body[:full].reshape(-1, 60)
lines[:, 60] = 10
with open(path, "wb") as fh:
    fh.write(b">chr_bench\n")
    fh.write(lines.tobytes())
    if body.size > full:
        fh.write(body[full:].tobytes() + b"\n")
del lines


def file_to_tsv():
    runner = RecordRunner(pipe)
    parts = []
    for kind, key, rows in runner.results(read_multi_fasta_device(path)):
        parts.append(rows_text_batch(path, key, rows) if kind == "batch" else rows_text(path, key, rows))
    return "".join(parts).encode()


try:
    file_to_tsv()
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = file_to_tsv()
        print(f"e2e {1e3 * (time.perf_counter() - t):.1f} ms, {len(out)} TSV bytes, reserved {torch.cuda.memory_reserved() >> 20} MiB", flush=True)
    pr = cProfile.Profile()
    pr.enable()
    file_to_tsv()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print(s.getvalue())
finally:
    os.unlink(path)
