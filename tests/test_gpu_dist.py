"""The RCCL ("nccl") code path of the record gather on a real GPU.  Only one GPU is available to the
tests, so the process group has a single rank: it still runs init_process_group("nccl"), barrier and
the all_gather calls on HBM tensors exactly as the N > 1 launch does (the N = 2 logic itself is covered
on CPU with gloo in tests/test_host.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _worker(q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    from deepgrp_amd.distributed import gather_records
    from deepgrp_amd.pipeline import SEGMENT_DTYPE
    rows = np.zeros(5, SEGMENT_DTYPE)
    rows["start"] = [50, 10, 30, 5, 7]
    rows["end"] = rows["start"] + 3
    rows["label"] = [1, 2, 3, 4, 1]
    rows["contig"] = [1, 1, 0, 0, 2]
    out = gather_records(rows, dev)
    empty = gather_records(np.zeros(0, SEGMENT_DTYPE), dev)
    t = torch.ones(1, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    q.put((out.tolist(), len(empty), float(t.item())))
    dist.destroy_process_group()


def test_gather_records_over_rccl_single_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(q,))
    p.start()
    out, nempty, one = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert [r[3] for r in out] == [0, 0, 1, 1, 2] and [r[0] for r in out] == [5, 30, 10, 50, 7]
    assert nempty == 0 and one == 1.0


def _split_worker(rank, world, port, fasta, model, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      DGRP_DIST_BACKEND="gloo")
    from deepgrp_amd.__main__ import main
    main(["-b", "7", "predict", model, fasta, "--output", out, "--split_contigs"])


@pytest.mark.parametrize("kind,world", [("gru128", 2), ("attention_defaults", 2), ("attention_defaults", 3)])
def test_split_contigs_ranks_on_one_gpu(tmp_path, kind, world):
    """--split_contigs (distributed.run_split) with 2 or 3 ranks (all on GPU 0, gloo as the transport) gives the same TSV,
    byte for byte, as a single-process run: window shares spill T - step rows into the next rank's rows and max-combine
    exactly, the short last batch (-b 7: SURVEY Q2) is placed by every rank, scores travel as float32 + int8.  Both the
    benchmark's model and a model of the reference's default shape (defaults.toml: 60 units, window 342, attention)."""
    import torch.multiprocessing as mp
    from deepgrp_amd import model as dgmodel, synthetic
    from deepgrp_amd.__main__ import main
    mpath = str(tmp_path / "m.hdf5")
    if kind == "gru128":
        w = synthetic.trained_weights()
        dgmodel.save_keras_hdf5(mpath, w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, vecsize=200)
    else:
        w = synthetic.synthetic_weights(60, 5, attention=True, seed=5, gain=2.0)
        dgmodel.save_keras_hdf5(mpath, w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], w["scale"], vecsize=342)
    fa = tmp_path / "two.fa"
    with open(fa, "wb") as fh:
        for k, n in enumerate((300_000, 123_457, 500, 2_000)):       # the last two: fewer windows than ranks x 16 / than one batch
            raw = synthetic.synthetic_chromosome(n, contig=k, flank=100)
            fh.write(b">rec%d\n" % k + b"\n".join(raw[i:i + 70] for i in range(0, len(raw), 70)) + b"\n")
    single = str(tmp_path / "single.tsv")
    main(["-b", "7", "predict", mpath, str(fa), "--output", single])
    split = str(tmp_path / "split.tsv")
    ctx = mp.get_context("spawn")
    port = 29800 + os.getpid() % 150
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, str(fa), mpath, split)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert open(split).read() == open(single).read()
    assert open(single).read().count("\n") > 10


def _shard_worker(rank, world, port, fasta, model, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      DGRP_DIST_BACKEND="gloo")
    import json
    from deepgrp_amd.__main__ import CommandLineParser, main
    main(["-b", "7", "predict", model] + fasta.split(",") + ["--output", out])
    with open(f"{out}.rank{rank}.json", "w") as fh:
        json.dump(CommandLineParser.last_sharded, fh)


def test_contig_sharding_two_ranks_many_records(tmp_path):
    """Contig sharding with 2 ranks (both on GPU 0, gloo transport) over a file of 300 short and 2 longer records plus a second
    file with records that need the reference's line loop: every rank ingests ONLY its byte ranges (about half of the bytes each,
    all of them once between the two), batches its share (dgrp_predict_batch), rank 0 gathers; TSV byte-identical to the
    single-process run."""
    import torch.multiprocessing as mp
    from deepgrp_amd import model as dgmodel, synthetic
    from deepgrp_amd.__main__ import main
    w = synthetic.trained_weights()
    mpath = str(tmp_path / "m.hdf5")
    dgmodel.save_keras_hdf5(mpath, w["kernel"], w["recurrent_kernel"], w["bias"], w["ff_kernel"], w["ff_bias"], None, vecsize=200)
    raw = synthetic.synthetic_chromosome(1_200_000, contig=3, flank=1000)[5000:-5000]
    import numpy as np
    rng = np.random.default_rng(1)
    fa = tmp_path / "many.fa"
    pos = 0
    with open(fa, "wb") as fh:
        for k in range(302):
            n = 300_000 if k in (17, 200) else int(rng.integers(1, 3000))
            seq = raw[pos:pos + n]
            pos += n
            fh.write(b">r%d\n" % k + b"\n".join(seq[i:i + 70] for i in range(0, len(seq), 70)) + b"\n")
    odd = tmp_path / "odd.fa"
    body = raw[pos:pos + 40_000]
    with open(odd, "wb") as fh:
        fh.write(b"ACGT\nheaderless lines are dropped\n")                                       # a first chunk without '>'
        fh.write(b">crlf\r\n" + b"\r\n".join(body[i:i + 60] for i in range(0, 9000, 60)) + b"\r\n")
        fh.write(b">spaces inside\n" + b"\n".join(body[i:i + 50] + b"  " for i in range(9000, 15000, 50)) + b"\n")   # reference loop
        fh.write(b">lower\n" + body[15000:30000].lower() + b"\n>\nACGTACGT\n>empty\n>last\n" + body[30000:] + b"\n")
    single = str(tmp_path / "single.tsv")
    main(["-b", "7", "predict", mpath, str(fa), str(odd), "--output", single])
    sharded = str(tmp_path / "sharded.tsv")
    ctx = mp.get_context("spawn")
    port = 29600 + os.getpid() % 150
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, f"{fa},{odd}", mpath, sharded)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    assert open(sharded).read() == open(single).read()
    assert open(single).read().count("\n") > 300 and "\tlast\t" in open(single).read()
    import json
    stats = [json.load(open(f"{sharded}.rank{r}.json")) for r in range(2)]
    total = os.path.getsize(fa) + os.path.getsize(odd)
    assert stats[0]["file_bytes"] == total and stats[0]["uploaded_bytes"] + stats[1]["uploaded_bytes"] == total
    assert abs(stats[0]["uploaded_bytes"] - total / 2) < 0.15 * total                          # bytes uploaded per rank ~ file / N
    assert stats[0]["records"] + stats[1]["records"] == 302 + 5



@pytest.mark.parametrize("ranks", [1, 2])
def test_bench_line_one_and_two_ranks(ranks):
    """bench.py as the driver launches it -- alone for N = 1, under torch.distributed.run for N > 1 (here 2 ranks on GPU 0 with
    gloo as the transport) -- with its default legs on: ONE JSON line from rank 0, every rank leaves, within minutes.  (r02: the
    informational legs once ran a collective on rank 0 alone and an N = 2 launch never returned.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["--mbp", "2", "--steps", "2", "--warmup", "1", "--cpu-sample-bp", "20000", "--accuracy-windows", "64"]
    if ranks == 1:
        cmd = [sys.executable, "bench.py", "--gpus", "1"] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(29900 + os.getpid() % 90), "bench.py", "--gpus", str(ranks), "--backend", "gloo"] + args
    res = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=420)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["steps"] == 2 and out["value"] > 0 and out["unit"] == "Mbp/s"
    assert out["roofline"]["bound"] == "mfma" and 0 < out["roofline"]["frac"] < 1
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["kind"] == "port"           # at every N (rank 0, after the group is gone)
    assert out["roofline"]["traffic_source"] is None or "STORED" in out["roofline"]["traffic_source"]
    assert ("e2e" in out) == (ranks == 1)
    if ranks > 1:
        # what every rank saw: its GPU, its times -- and the sharded command line over ONE file of `ranks` records
        assert [r["rank"] for r in out["ranks"]] == list(range(ranks)) and all(r["kernel_ms"] > 0 and r["mbp_per_s"] > 0 for r in out["ranks"])
        assert all(r["pci_bus_id"] or r["uuid"] for r in out["ranks"]) and out["distinct_gpus"] == 1     # (both ranks on GPU 0 here)
        sf = out["sharded_file"]
        assert sf["records"] == ranks and sf["value"] > 0 and sf["rows_out"] > 0
        assert sum(sf["uploaded_bytes_per_rank"]) == sf["file_bytes"]
        assert all(abs(b - sf["file_bytes"] / ranks) < 0.1 * sf["file_bytes"] for b in sf["uploaded_bytes_per_rank"])
        assert sf["records_per_rank"] == [1] * ranks
