#!/usr/bin/env python3
"""Headline benchmark: Mbp/s predicted at window=200 stride=50, 5 classes (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mbp M]

A "step" is one pass of the whole hot path (class-index encode -> GRU + dense + softmax +
max-merge -> score transform -> MSS + vote -> segment extraction -> record gather) over one
synthetic chromosome per GPU whose bytes are already resident in HBM.  N > 1 is launched by
torchrun, one rank per GPU; records shard by contig (weak scaling, no data-path collective),
the only exchange is the gather of segment records to rank 0 over RCCL.

Prints ONE JSON line (rank 0) carrying `roofline` for the dominant kernel (the fused GRU
kernel, MFMA-bound, timed live with HIP events on its stream) and `cpu_baseline` (the CPU
restatement of the same path, oracle/, timed on this host on a bounded sample).

The timed path is the package's default for this model: the split-operand GRU kernel, whose class probabilities agree
with an fp32 evaluation to ~1e-6 on every base (`accuracy`).  `--fast` times the fp16-operand kernel instead (2.5x
faster; within 1e-3 of fp32 on 99.98 % of the bases of this workload, not on all); without the flag that mode is
measured after the timed region and reported as `fast_mode`, for information.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deepgrp_amd import synthetic                                  # noqa: E402
from deepgrp_amd.distributed import gather_records                 # noqa: E402
from deepgrp_amd.pipeline import ContigPipeline, DeviceModel, upload_sequence   # noqa: E402

T, STEP, UNITS, CLASSES, BATCH, MIN_MSS, XDROP = 200, 50, 128, 5, 256, 50, 50
MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md, chip-level parameters
FLOP_PER_WINDOW = 12 * UNITS * UNITS * T + 2 * UNITS * CLASSES * T


def cpu_baseline(weights, sample_bp: int):
    """The oracle (CPU restatement, all host threads) on a bounded sample of the same workload."""
    from oracle import oracle as orc
    seq = synthetic.synthetic_chromosome(sample_bp, contig=0, flank=1000).decode()
    w = orc.Weights(weights["kernel"], weights["recurrent_kernel"], weights["bias"], weights["ff_kernel"],
                    weights["ff_bias"], weights["scale"], T)
    threads = orc.lib().orc_max_threads()
    sample_bp = min(sample_bp, max(50_000, 25_000 * threads))     # ~10-30 s of wall clock whatever the core count

    def factory(idx):
        return lambda w0, nw: orc.nn_forward(idx, w, STEP, w0, nw, np.float32, threads)

    t0 = time.perf_counter()
    rows = orc.predict_contig(seq, factory, T, CLASSES, STEP, BATCH, MIN_MSS, XDROP, True)
    dt = time.perf_counter() - t0
    return {"value": round(sample_bp / dt / 1e6, 5), "unit": "Mbp/s", "cores": int(threads), "kind": "port",
            "sample": f"{sample_bp} bp synthetic contig, full path (oracle/dgrp_oracle.c, float32, OpenMP over windows), "
                      f"{dt:.1f} s, {len(rows)} rows"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mbp", type=float, default=50.0, help="chromosome size per GPU in Mbp (configs[1] = 50)")
    ap.add_argument("--weights", choices=("trained", "random"), default="trained",
                    help="trained: deepgrp_amd/data/synthetic_gru128.npz (fitted to the planted repeats, genome-like output); "
                         "random: Keras initialisers scaled by --gain (stationary noise, the MSS worst case)")
    ap.add_argument("--gain", type=float, default=3.0, help="weight scale of the random model")
    ap.add_argument("--cpu-sample-bp", type=int, default=400_000)
    ap.add_argument("--fast", action="store_true", help="time the fp16-operand GRU kernel instead of the default split-operand one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--accuracy-windows", type=int, default=4096, help="windows compared with the fp32 yardstick after the run (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
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
    from deepgrp_amd._lib import check, lib
    from deepgrp_amd.pipeline import stream_ptr

    def step():
        check(lib().dgrp_encode(d_seq.data_ptr(), n, d_idx.data_ptr(), stream_ptr()), "dgrp_encode")
        rows = pipe.run_idx(d_idx, startpos, contig=rank)
        return gather_records(rows, dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    pipe.event_log = []
    t0 = time.perf_counter()
    nrows = 0
    for _ in range(args.steps):
        nrows = len(step())
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    main_log = pipe.event_log
    # for information: the other fused kernel of this model on the same input, same step function, outside the timed region
    other = None
    if rank == 0 and model.supports_split:
        pipe_o = ContigPipeline(model, STEP, BATCH, MIN_MSS, XDROP, use_mss=True, fast=not args.fast)
        step_o = lambda: (check(lib().dgrp_encode(d_seq.data_ptr(), n, d_idx.data_ptr(), stream_ptr()), "dgrp_encode"),
                          pipe_o.run_idx(d_idx, startpos, contig=rank))[1]
        step_o()
        torch.cuda.synchronize()
        pipe_o.event_log = []
        t1 = time.perf_counter()
        for _ in range(max(1, min(args.steps, 3))):
            rows_o = step_o()
        torch.cuda.synchronize()
        dt_o = (time.perf_counter() - t1) / max(1, min(args.steps, 3))
        other = {"value": round(n_bases / dt_o / 1e6, 3), "unit": "Mbp/s (1 GPU, this rank)", "ms_per_step": round(dt_o * 1e3, 3),
                 "kernel_ms": round(float(np.mean([a.elapsed_time(b) for a, b, _ in pipe_o.event_log])), 3), "rows_out": int(len(rows_o))}
    # dominant kernel: the fused GRU kernel (one launch per dgrp_forward_merge for this model)
    pipe.event_log = main_log
    kern_ms = [a.elapsed_time(b) for a, b, _ in pipe.event_log]
    kern_windows = [w for _, _, w in pipe.event_log]
    avg_ms = float(np.mean(kern_ms))
    achieved = float(np.mean(kern_windows)) * FLOP_PER_WINDOW / (avg_ms * 1e-3) / 1e12

    # HBM traffic of that kernel from the PMC passes of profiles/ (separate rocprofv3 --pmc runs of this
    # command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), scaled to this launch size
    fused_name = "gru_fused_kernel<4, 0, %s>" % ("true" if model.kernel_flags & 1 else "false")
    split_name = "gru_split2_kernel<0>"          # 128-unit class, single-record launch: two row tiles per wave (gru_kernel.hip)
    kernel_name = fused_name if not pipe.split else split_name
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_gru_traffic.json")
    if os.path.exists(tpath) and args.weights == "trained":
        with open(tpath) as fh:
            per_kernel = json.load(fh).get("kernels", {}).get(kernel_name.split("<")[0])
        if per_kernel:
            traffic = round(per_kernel["hbm_bytes_per_window"] * float(np.mean(kern_windows)))
    if rank == 0:
        value = n_bases * world * args.steps / dt / 1e6
        out = {
            "metric": "Mbp/sec predicted (whole node) at window=200 stride=50, 5-class",
            "value": round(value, 3), "unit": "Mbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16",
            "dtype_detail": ("GRU: f16 MFMA operands, f32 accumulate and state" if not pipe.split else
                             "GRU: weights and hidden state as f16 hi+lo pairs on the MFMA (three passes, the 2^-22 cross term dropped), "
                             "f32 accumulate and state: pre-activations to fp32 rounding") + "; post-processing f32/f64/int exactly as the reference",
            "data": "synthetic",
            "config": {"workload": f"{args.mbp:g} Mbp synthetic chromosome per GPU (BASELINE configs[1]), "
                                   f"window={T} stride={STEP} hidden={UNITS} {CLASSES}-class, batch_size={BATCH}, "
                                   f"MSS min_len={MIN_MSS} xdrop={XDROP}, {wdesc}",
                       "rows_out": int(nrows), "parallelism": f"contig-sharded x{world}"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F16_DENSE_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / MFMA_F16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                         "kernel": kernel_name, "flop_per_window": FLOP_PER_WINDOW, "avg_launch_ms": round(avg_ms, 3),
                         "windows_per_launch": int(np.mean(kern_windows)), "launches_timed": len(kern_ms)},
        }
        # how far each fused kernel is from a plain-fp32 evaluation of the same model on the device (ref_kernels.hip), on
        # windows spread over this rank's chromosome; outside the timed region
        def acc_obj(acc):
            return {"windows": acc["windows_checked"], "median_window_max_abs_dp": round(acc["median_window_max"], 8),
                    "q99_window_max_abs_dp": round(acc["q99_window_max"], 8), "max_abs_dp": round(acc["max_abs_diff"], 8),
                    "frac_positions_above_1e-3": round(acc["positions_above_1e-3"] / acc["positions_checked"], 7),
                    "argmax_flips": acc["argmax_flips"], "positions": acc["positions_checked"]}
        if args.accuracy_windows > 0:
            out["accuracy"] = dict(yardstick="plain-fp32 HIP kernels (dgrp_forward_windows_reference), themselves within 2e-5 of the float64 CPU statement",
                                   kernel=kernel_name, **acc_obj(model.check_accuracy(d_idx, STEP, args.accuracy_windows, level=1 if pipe.split else 0)))
        if other is not None:
            other["kernel"] = fused_name if pipe.split else split_name
            if args.accuracy_windows > 0:
                other["accuracy"] = acc_obj(model.check_accuracy(d_idx, STEP, args.accuracy_windows, level=0 if pipe.split else 1))
            out["fast_mode" if pipe.split else "default_mode"] = other
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(weights, args.cpu_sample_bp)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
