#!/usr/bin/env python3
"""Headline benchmark: Mbp/s predicted at window=200 stride=50, 5 classes (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mbp M]

Workload: one synthetic chromosome of M Mbp per GPU, default 250 (BASELINE configs[2], the size the north-star target is
quoted on; `--mbp 50` = configs[1]).  A "step" is one pass of the whole hot path over it -- class-index encode -> GRU +
dense + softmax + max-merge -> score transform -> MSS + vote -> segment extraction -> record gather -- through the
package's default entry point (`dgrp_predict_record`, the call the command line makes per record), with the sequence
bytes already resident in HBM.  N > 1: one rank per GPU (`python bench.py --gpus N` starts torch.distributed.run itself
when it was not started by it), records shard by contig (weak scaling, no data-path collective), the only exchange is
the gather of segment records to rank 0 over RCCL.

Prints ONE JSON line (rank 0):
  value / ms_per_step   the device-resident step above, K steps between barriers, max over ranks
  roofline              the recurrent kernel (MFMA-bound): algorithmic flop of a launch / its duration, from HIP events the
                        library records around the launch on its stream INSIDE the timed steps (dgrp_kernel_timer_*)
  stages                per-stage milliseconds of one step (a separate, staged pass over the same input)
  e2e                   the same record from FASTA bytes in host memory (a file in /dev/shm) to TSV bytes in host memory:
                        device ingest (A1) + upload over PCIe + the step + TSV text (A12), as the command line runs it
  accuracy              the timed kernel against the plain-fp32 kernels on windows spread over the chromosome
  fast_mode             for information: the fp16-operand kernel (`--fast`) on the same input, and its accuracy
  cpu_baseline          the CPU restatement of the same path (oracle/) on bounded samples: all host threads, and one thread
                        (the reference's default --threads 1); rank 0, at N > 1 after the process group is gone
  ranks                 N > 1: what every rank saw -- its GPU (PCI bus id, uuid), its step and kernel times, its rows (all_gather)
  sharded_file          N > 1: ONE FASTA file of N records through the command line's sharded path (deepgrp_amd/__main__.py:
                        chunk table from per-rank host scans, records shared out by byte length, rank-local upload + ingest,
                        RCCL gather of the rows, rank 0 writes the TSV), with the bytes every rank uploaded
(stages, e2e, accuracy and fast_mode at N = 1 only.)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T, STEP, UNITS, CLASSES, BATCH, MIN_MSS, XDROP = 200, 50, 128, 5, 256, 50, 50
MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md, chip-level parameters
FLOP_PER_WINDOW = 12 * UNITS * UNITS * T + 2 * UNITS * CLASSES * T


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mbp", type=float, default=250.0, help="chromosome size per GPU in Mbp (250 = configs[2], 50 = configs[1])")
    ap.add_argument("--weights", choices=("trained", "random"), default="trained",
                    help="trained: deepgrp_amd/data/synthetic_gru128.npz (fitted to the planted repeats, genome-like output); "
                         "random: Keras initialisers scaled by --gain (stationary noise, the MSS worst case)")
    ap.add_argument("--gain", type=float, default=3.0, help="weight scale of the random model")
    ap.add_argument("--cpu-sample-bp", type=int, default=2_000_000, help="bases of the all-threads CPU sample (10-30 s of CPU work on a 128-thread host)")
    ap.add_argument("--fast", action="store_true", help="time the fp16-operand GRU kernel instead of the default split-operand one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed steps: no stages / e2e / other-mode / accuracy legs")
    ap.add_argument("--accuracy-windows", type=int, default=4096, help="windows compared with the fp32 yardstick after the run (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--master-port", type=int, default=29533, help="rendezvous port when bench.py starts the ranks itself")
    ap.add_argument("--no-sharded-file", action="store_true", help="N > 1: skip the sharded command-line leg")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as fresh child processes (nothing in this process has
    touched the GPU yet) and pass their output through."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def cpu_baseline(weights, sample_bp: int):
    """The oracle (CPU restatement, all host threads) on a bounded sample of the same workload."""
    import numpy as np
    from deepgrp_amd import synthetic
    from oracle import oracle as orc
    w = orc.Weights(weights["kernel"], weights["recurrent_kernel"], weights["bias"], weights["ff_kernel"],
                    weights["ff_bias"], weights["scale"], T)
    threads = orc.lib().orc_max_threads()
    sample_bp = min(sample_bp, max(50_000, 25_000 * threads))     # a few seconds of wall clock whatever the core count

    def run(bp, nthreads):
        seq = synthetic.synthetic_chromosome(bp, contig=0, flank=1000).decode()

        def factory(idx):
            return lambda w0, nw: orc.nn_forward(idx, w, STEP, w0, nw, np.float32, nthreads)
        t0 = time.perf_counter()
        rows = orc.predict_contig(seq, factory, T, CLASSES, STEP, BATCH, MIN_MSS, XDROP, True)
        return time.perf_counter() - t0, len(rows)

    dt, nrows = run(sample_bp, threads)
    # the reference's default is --threads 1 (deepgrp/__main__.py:133-139): the same path on one thread, a sample of a few seconds
    one_bp = min(300_000, max(60_000, sample_bp // 6))
    dt1, _ = run(one_bp, 1)
    return {"value": round(sample_bp / dt / 1e6, 5), "unit": "Mbp/s", "cores": int(threads), "kind": "port",
            "sample": f"{sample_bp} bp synthetic contig, full path (oracle/dgrp_oracle.c, float32, OpenMP over windows), "
                      f"{dt:.1f} s, {nrows} rows",
            "one_thread": {"value": round(one_bp / dt1 / 1e6, 6), "unit": "Mbp/s", "sample": f"{one_bp} bp, {dt1:.1f} s"}}


def write_fasta(path, header: bytes, raw: bytes) -> None:
    """One record, 60-column lines."""
    import numpy as np
    body = np.frombuffer(raw, np.uint8)
    full = body.size // 60 * 60
    lines = np.empty((full // 60, 61), np.uint8)
    lines[:, :60] = body[:full].reshape(-1, 60)
    lines[:, 60] = 10
    with open(path, "wb") as fh:
        fh.write(b">" + header + b"\n")
        fh.write(lines.tobytes())
        if body.size > full:
            fh.write(body[full:].tobytes() + b"\n")


def gpu_identity(torch, dev) -> dict:
    """What tells two GPUs apart: PCI bus id and uuid of the device this rank runs on."""
    prop = torch.cuda.get_device_properties(dev)
    out = {"device": int(dev.index), "name": prop.name}
    try:
        out["pci_bus_id"] = "%04x:%02x:%02x.0" % (prop.pci_domain_id, prop.pci_bus_id, prop.pci_device_id)
    except AttributeError:
        out["pci_bus_id"] = None
    try:
        out["uuid"] = str(prop.uuid)
    except AttributeError:
        out["uuid"] = None
    return out


def sharded_file_leg(args, rank, world, raw, weights, n_bases, dist, torch):
    """ONE FASTA file holding every rank's chromosome, through `deepgrp predict` in-process on all ranks (the sharded path of
    deepgrp_amd/__main__.py): per-rank chunk scan -> LPT by byte length -> rank-local upload + device ingest -> the step ->
    record gather -> rank 0 writes the TSV.  Timed between barriers on the second of two runs."""
    from deepgrp_amd import fasta as dgfasta, model as dgmodel
    from deepgrp_amd.__main__ import CommandLineParser, main as cli_main
    # room for every rank's part and the joined file (a container's /dev/shm may be 64 MB): decided together, so that no rank is left
    # waiting in a barrier for one that could not write
    need = (2 * world + 1) * (len(raw) + len(raw) // 60 + 4096)
    def room(d):
        try:
            st = os.statvfs(d)
            return os.path.isdir(d) and os.access(d, os.W_OK) and st.f_bavail * st.f_frsize > need
        except OSError:
            return False
    choice = [None] * world
    dist.all_gather_object(choice, "/dev/shm" if room("/dev/shm") else "/tmp" if room("/tmp") else "")
    if any(c != choice[0] for c in choice) or not choice[0]:
        return {"skipped": f"no directory with {need >> 20} MB free on every rank (/dev/shm, /tmp)"} if rank == 0 else None
    shm = choice[0]
    tag = os.environ.get("MASTER_PORT", "0")
    part = os.path.join(shm, f"dgrp_bench_{tag}_part{rank}.fa")
    fa_path = os.path.join(shm, f"dgrp_bench_{tag}_all.fa")
    mpath = os.path.join(shm, f"dgrp_bench_{tag}_model.hdf5")
    tsv = os.path.join(shm, f"dgrp_bench_{tag}_out.tsv")
    def together(err):                                       # an I/O error on one rank reaches every rank (then: leg skipped)
        errs = [None] * world
        dist.all_gather_object(errs, err)
        return next((e for e in errs if e), None)
    try:
        err = None
        try:
            write_fasta(part, b"chr_rank%d" % rank, raw)
        except OSError as e:
            err = f"rank {rank}: {e}"
        err = together(err)
        if err is None and rank == 0:
            try:
                import shutil
                with open(fa_path, "wb") as dst:
                    for r in range(world):
                        with open(os.path.join(shm, f"dgrp_bench_{tag}_part{r}.fa"), "rb") as src:
                            shutil.copyfileobj(src, dst, 16 << 20)
                dgmodel.save_keras_hdf5(mpath, weights["kernel"], weights["recurrent_kernel"], weights["bias"], weights["ff_kernel"],
                                        weights["ff_bias"], weights["scale"], vecsize=T)
            except OSError as e:
                err = f"rank 0: {e}"
        err = together(err)
        if os.path.exists(part):
            os.unlink(part)
        if err is not None:
            return {"skipped": err} if rank == 0 else None
        argv = ["-b", str(BATCH), "-s", str(STEP), "-x", str(XDROP), "-l", str(MIN_MSS), "predict", mpath, fa_path, "--output", tsv]
        if args.fast:
            argv.append("--fast")
        times = []
        for _ in range(2):
            up0 = dgfasta.UPLOAD_STATS["bytes"]
            dist.barrier()
            torch.cuda.synchronize()
            t = time.perf_counter()
            cli_main(argv)                                  # ends with a barrier of its own (rank 0 has written the file)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t)
        stats = [None] * world
        dist.all_gather_object(stats, {"rank": rank, "uploaded_bytes": dgfasta.UPLOAD_STATS["bytes"] - up0,
                                       "records": CommandLineParser.last_sharded.get("records"), "s": round(times[-1], 4)})
        if rank != 0:
            return None
        file_bytes = os.path.getsize(fa_path)
        dt = max(x["s"] for x in stats)
        with open(tsv, "rb") as fh:
            rows_out = fh.read().count(b"\n")
        return {"value": round(n_bases * world / dt / 1e6, 3), "unit": "Mbp/s (whole job)", "ms": round(dt * 1e3, 3),
                "file_bytes": int(file_bytes), "records": world, "rows_out": int(rows_out),
                "uploaded_bytes_per_rank": [x["uploaded_bytes"] for x in stats], "records_per_rank": [x["records"] for x in stats],
                "what": "deepgrp predict <model.hdf5> <one FASTA file of N records> --output <tsv> on N ranks: model load, per-rank chunk "
                        "scan, rank-local upload + device ingest of the rank's byte ranges only, the step, RCCL gather of the rows, "
                        "TSV written by rank 0; second of two runs"}
    finally:
        dist.barrier()
        if rank == 0:
            for f in (fa_path, mpath, tsv):
                if os.path.exists(f):
                    os.unlink(f)
        if os.path.exists(part):
            os.unlink(part)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))                                # before any GPU call in this process
    import ctypes as C

    import numpy as np
    import torch
    import torch.distributed as dist

    from deepgrp_amd import synthetic
    from deepgrp_amd._lib import check, lib
    from deepgrp_amd.distributed import gather_records
    from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, stream_ptr, upload_sequence

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl":
        sys.exit(f"rank {rank}: local rank {local_rank} but only {ndev} GPUs visible")
    local_dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    if args.weights == "trained":
        weights = synthetic.trained_weights()
        wdesc = "synthetic model fitted to the planted tandem repeats (tools/train_synth_model.py)"
    else:
        weights = synthetic.synthetic_weights(UNITS, CLASSES, attention=False, seed=7, gain=args.gain)
        wdesc = f"random-init weights gain={args.gain:g}"
    model = DeviceModel(weights["kernel"], weights["recurrent_kernel"], weights["bias"], weights["ff_kernel"],
                        weights["ff_bias"], weights["scale"], vecsize=T)
    pipe = ContigPipeline(model, STEP, BATCH, MIN_MSS, XDROP, use_mss=True, fast=args.fast)
    n_bases = int(args.mbp * 1e6)
    raw = synthetic.synthetic_chromosome(n_bases, contig=rank)
    startpos, d_idx = upload_sequence(raw)                 # also validates the encoder once
    # the resident input of a step: the stripped sequence bytes in HBM
    d_seq = torch.from_numpy(np.frombuffer(raw, np.uint8)[startpos:startpos + d_idx.numel()].copy()).to(dev)
    n = d_seq.numel()
    L = lib()

    def step(p=pipe):
        check(L.dgrp_encode(d_seq.data_ptr(), n, d_idx.data_ptr(), stream_ptr()), "dgrp_encode")
        rows = p.run_idx(d_idx, startpos, contig=rank)     # dgrp_predict_record: the package's default path
        local_rows[0] = len(rows)
        return gather_records(rows, dev)

    local_rows = [0]                                       # rows of THIS rank's record (the gather returns every rank's on rank 0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def kernel_timer_read():
        ms, launches, windows = C.c_double(0), C.c_int64(0), C.c_int64(0)
        check(L.dgrp_kernel_timer_read(C.byref(ms), C.byref(launches), C.byref(windows)), "dgrp_kernel_timer_read")
        return ms.value, launches.value, windows.value

    for _ in range(args.warmup):
        step()
    fence()
    check(L.dgrp_kernel_timer_enable(1), "dgrp_kernel_timer_enable")      # HIP events around the recurrent kernel, on its stream
    t0 = time.perf_counter()
    nrows = 0
    for _ in range(args.steps):
        nrows = len(step())
    fence()
    dt = dt_local = time.perf_counter() - t0
    kern_ms, kern_launches, kern_windows = kernel_timer_read()
    check(L.dgrp_kernel_timer_enable(0), "dgrp_kernel_timer_enable")
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    avg_ms = kern_ms / max(kern_launches, 1)
    win_per_launch = kern_windows / max(kern_launches, 1)
    achieved = win_per_launch * FLOP_PER_WINDOW / (avg_ms * 1e-3) / 1e12

    # the informational legs run on a lone rank only: they are not part of the contract's line, the other ranks would sit in the final
    # barrier meanwhile, and a leg that reached a collective (the record gather inside step()) on rank 0 alone would never return
    extras = rank == 0 and world == 1 and not args.no_extras
    stages = e2e = other = None
    if extras:
        # ---- per-stage milliseconds of one step: the staged form of the same path, a stage at a time
        def staged():
            out = {}
            torch.cuda.synchronize()
            t = time.perf_counter()
            check(L.dgrp_encode(d_seq.data_ptr(), n, d_idx.data_ptr(), stream_ptr()), "dgrp_encode")
            torch.cuda.synchronize(); out["encode"] = time.perf_counter() - t; t = time.perf_counter()
            merged = pipe.merged(d_idx)
            torch.cuda.synchronize(); out["forward_merge"] = time.perf_counter() - t; t = time.perf_counter()
            labels = pipe.labels(merged)
            torch.cuda.synchronize(); out["scores_mss_vote"] = time.perf_counter() - t; t = time.perf_counter()
            del merged
            rows = pipe.segments(labels, startpos, rank)
            out["segments_and_readback"] = time.perf_counter() - t
            return out, len(rows)
        staged()
        reps = [staged() for _ in range(2)]
        assert all(r[1] == local_rows[0] for r in reps), "staged path and dgrp_predict_record disagree on the row count"
        stages = {k: round(float(np.mean([r[0][k] for r in reps])) * 1e3, 3) for k in reps[0][0]}

        # ---- end to end: FASTA bytes in host memory -> TSV bytes in host memory, as the command line runs a file
        from deepgrp_amd.fasta import read_multi_fasta_device
        from deepgrp_amd.runner import RecordRunner, rows_text, rows_text_batch
        shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
        fa_path = os.path.join(shm, f"dgrp_bench_{os.getpid()}.fa")
        try:
            write_fasta(fa_path, b"chr_bench", raw)
            fasta_bytes = os.path.getsize(fa_path)

            def file_to_tsv():
                runner = RecordRunner(pipe)
                parts = []
                for kind, key, rows in runner.results(read_multi_fasta_device(fa_path)):
                    parts.append(rows_text_batch(fa_path, key, rows) if kind == "batch" else rows_text(fa_path, key, rows))
                return "".join(parts).encode()
            file_to_tsv()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t = time.perf_counter()
                tsv = file_to_tsv()
                ts.append(time.perf_counter() - t)
            e2e = {"value": round(n_bases / float(np.mean(ts)) / 1e6, 3), "unit": "Mbp/s (1 GPU, this rank)",
                   "ms": round(float(np.mean(ts)) * 1e3, 3), "fasta_bytes": int(fasta_bytes), "tsv_bytes": len(tsv),
                   "rows_out": tsv.count(b"\n"),
                   "what": "FASTA file in host memory (60-column lines) -> device ingest (A1) incl. upload over PCIe -> the step -> TSV "
                           "text in host memory (A12); model already loaded"}
        finally:
            if os.path.exists(fa_path):
                os.unlink(fa_path)

        # ---- for information: the other fused kernel of this model on the same input, same step function
        if model.supports_split:
            pipe_o = ContigPipeline(model, STEP, BATCH, MIN_MSS, XDROP, use_mss=True, fast=not args.fast)
            step(pipe_o)
            torch.cuda.synchronize()
            check(L.dgrp_kernel_timer_enable(1), "dgrp_kernel_timer_enable")
            t1 = time.perf_counter()
            k_o = max(1, min(args.steps, 3))
            for _ in range(k_o):
                rows_o = step(pipe_o)
            torch.cuda.synchronize()
            dt_o = (time.perf_counter() - t1) / k_o
            ms_o, launches_o, _w = kernel_timer_read()
            check(L.dgrp_kernel_timer_enable(0), "dgrp_kernel_timer_enable")
            other = {"value": round(n_bases / dt_o / 1e6, 3), "unit": "Mbp/s (1 GPU, this rank)", "ms_per_step": round(dt_o * 1e3, 3),
                     "kernel_ms": round(ms_o / max(launches_o, 1), 3), "rows_out": int(len(rows_o))}

    # ---- N > 1: what every rank saw (the evidence that N distinct GPUs did the work), then the sharded command-line leg
    ranks_block = sharded = None
    if world > 1:
        me = dict(rank=rank, **gpu_identity(torch, dev), ms_per_step=round(dt_local / args.steps * 1e3, 3),
                  mbp_per_s=round(n_bases * args.steps / dt_local / 1e6, 3), kernel_ms=round(avg_ms, 3),
                  kernel_launches=int(kern_launches), rows=int(local_rows[0]), host=os.uname().nodename, pid=os.getpid())
        ranks_block = [None] * world
        dist.all_gather_object(ranks_block, me)
        if not args.no_sharded_file:
            sharded = sharded_file_leg(args, rank, world, raw, weights, n_bases, dist, torch)

    # HBM traffic of the timed kernel from the PMC passes of profiles/ (separate rocprofv3 --pmc runs of this command;
    # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), scaled to this launch size
    fused_name = "gru_fused_kernel<4, 0, %s>" % ("true" if model.kernel_flags & 1 else "false")
    split_name = "gru_split2_kernel<0, %s>" % ("true" if model.kernel_flags & 1 else "false")
    kernel_name = fused_name if not pipe.split else split_name
    traffic = traffic_source = None
    tfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_gru_traffic.json")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    if tfiles and args.weights == "trained":
        with open(os.path.join(ROOT, "profiles", tfiles[-1])) as fh:
            per_kernel = json.load(fh).get("kernels", {}).get(kernel_name.split("<")[0])
        if per_kernel:
            traffic = round(per_kernel["hbm_bytes_per_window"] * win_per_launch)
            traffic_source = (f"profiles/{tfiles[-1]}: a STORED per-window figure from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                              "command (2*FETCH + WRITE, gfx950 correction), scaled to this launch's windows -- not a counter of this run")
    if rank == 0:
        value = n_bases * world * args.steps / dt / 1e6
        cfg = "configs[2]" if abs(args.mbp - 250) < 1e-9 else "configs[1]" if abs(args.mbp - 50) < 1e-9 else "custom size"
        out = {
            "metric": "Mbp/sec predicted (whole node) at window=200 stride=50, 5-class",
            "value": round(value, 3), "unit": "Mbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16",
            "dtype_detail": ("GRU: f16 MFMA operands, f32 accumulate and state" if not pipe.split else
                             "GRU: weights and hidden state as f16 hi+lo pairs on the MFMA (three passes, the 2^-22 cross term dropped), "
                             "f32 accumulate and state: pre-activations to fp32 rounding") + "; post-processing f32/f64/int exactly as the reference",
            "data": "synthetic",
            "config": {"workload": f"{args.mbp:g} Mbp synthetic chromosome per GPU (BASELINE {cfg}), "
                                   f"window={T} stride={STEP} hidden={UNITS} {CLASSES}-class, batch_size={BATCH}, "
                                   f"MSS min_len={MIN_MSS} xdrop={XDROP}, {wdesc}; a step = dgrp_encode + dgrp_predict_record + record "
                                   f"gather on HBM-resident sequence bytes (FASTA ingest, upload and TSV text: see e2e)",
                       "rows_out": int(nrows), "parallelism": f"contig-sharded x{world}"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F16_DENSE_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / MFMA_F16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_source": traffic_source,
                         "kernel": kernel_name, "flop_per_window": FLOP_PER_WINDOW, "avg_launch_ms": round(avg_ms, 3),
                         "windows_per_launch": int(win_per_launch), "launches_timed": int(kern_launches),
                         "timing": "HIP events recorded by the library around each launch on its stream inside the timed steps (dgrp_kernel_timer_*)"},
        }
        if stages is not None:
            out["stages"] = dict(unit="ms per step (staged pass, one stage at a time, host clock around a device sync)", **stages)
        if e2e is not None:
            out["e2e"] = e2e

        # how far each fused kernel is from a plain-fp32 evaluation of the same model on the device (ref_kernels.hip), on
        # windows spread over this rank's chromosome; outside the timed region
        def acc_obj(acc):
            return {"windows": acc["windows_checked"], "median_window_max_abs_dp": round(acc["median_window_max"], 8),
                    "q99_window_max_abs_dp": round(acc["q99_window_max"], 8), "max_abs_dp": round(acc["max_abs_diff"], 8),
                    "frac_positions_above_1e-3": round(acc["positions_above_1e-3"] / acc["positions_checked"], 7),
                    "argmax_flips": acc["argmax_flips"], "positions": acc["positions_checked"]}
        if args.accuracy_windows > 0 and extras:
            out["accuracy"] = dict(yardstick="plain-fp32 HIP kernels (dgrp_forward_windows_reference), themselves within 2e-5 of the float64 CPU statement",
                                   kernel=kernel_name, **acc_obj(model.check_accuracy(d_idx, STEP, args.accuracy_windows, level=1 if pipe.split else 0)))
        if other is not None:
            other["kernel"] = fused_name if pipe.split else split_name
            if args.accuracy_windows > 0 and extras:
                other["accuracy"] = acc_obj(model.check_accuracy(d_idx, STEP, args.accuracy_windows, level=0 if pipe.split else 1))
            out["fast_mode" if pipe.split else "default_mode"] = other
        if ranks_block is not None:
            out["ranks"] = ranks_block
            out["distinct_gpus"] = len({(r["host"], r["pci_bus_id"] or r["uuid"] or r["device"]) for r in ranks_block})
        if sharded is not None:
            out["sharded_file"] = sharded
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the CPU leg needs no other rank and no collective: at N > 1 it runs after the process group is gone (the other ranks have
        # left; nobody waits in a barrier while the host cores are busy)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(weights, args.cpu_sample_bp)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
