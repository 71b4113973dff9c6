#!/usr/bin/env python3
"""deepgrp_amd command line -- drop-in for `deepgrp [flags] predict <model.hdf5> <FASTA>...`
(deepgrp/__main__.py:86-297 of the reference): same flags and defaults, same 5-column TSV.

Differences, all additive:
  * the README's short form `deepgrp <modelfile> <fastafile>` is accepted too;
  * `--xla` and `--threads` are accepted and ignored (there is no TensorFlow here);
  * under torchrun (WORLD_SIZE > 1) the records of all input files are sharded by contig over
    the GPUs and rank 0 writes the rows in input order;
  * `train` exits with an error: training is TensorFlow's job in the reference and out of scope.
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
from typing import Iterator, List, TextIO, Tuple

import numpy as np

logging.basicConfig()
_LOG = logging.getLogger(__name__)


def _read_multi_fasta(filestream: TextIO) -> Iterator[Tuple[str, str]]:
    """Reads a multi FASTA file (deepgrp/__main__.py:20-43): header = text after '>', sequence
    lines upper-cased and joined; a record without header is dropped; a blank line raises
    IndexError exactly like `line[0]` does in the reference.  (The loop itself: fasta.LineLoop.)"""
    from .fasta import read_multi_fasta_lines
    _LOG.debug("Reading FASTA file.")
    yield from read_multi_fasta_lines(filestream)


def _predict(dnasequence: str, model, options, step_size: int, use_mss: bool) -> Tuple[np.ndarray, int]:
    """Runs a prediction for one sequence (deepgrp/__main__.py:46-83): returns the label per base
    (int64, after MSS or the softmax path) and the number of leading N's."""
    from .pipeline import ContigPipeline, upload_sequence
    _LOG.debug("One hot encoding sequence.")
    start_pos, d_idx = upload_sequence(dnasequence.encode("utf-8"))
    _LOG.debug("Start prediction.")
    pipe = ContigPipeline(model, step_size, options.batch_size, options.min_mss_len, options.xdrop_len, use_mss)
    merged = pipe.merged(d_idx)
    _LOG.debug("Finish prediction.")
    if use_mss:
        _LOG.debug("Applying MSS.")
    labels = pipe.labels(merged)
    return labels.cpu().numpy().astype(np.int64), start_pos


class CommandLineParser:
    """Commandline parser (deepgrp/__main__.py:86-250)."""

    def __init__(self, **kwargs):
        kwargs.setdefault("prog", "deepgrp")
        kwargs.setdefault("formatter_class", argparse.ArgumentDefaultsHelpFormatter)
        kwargs.setdefault("description", "DeepGRP - Prediction of repetitive elements (MI355X / HIP)")
        self.parser = argparse.ArgumentParser(**kwargs)
        self.args = None
        self.threads = 1
        self.xla = False
        self.verbose = 0
        subparsers = self.parser.add_subparsers(help="sub-command help", dest="command")
        self.parser.add_argument("--batch_size", "-b", type=int, default=256,
                                 help="Batch size of the reference's TensorFlow loop; only its placement arithmetic matters here")
        self.parser.add_argument("--step_size", "-s", type=int, default=50, help="Window step size")
        self.parser.add_argument("--xdrop_length", "-x", type=int, default=50,
                                 help="XDrop parameter for MSS algorithm, ignored if --no_use_mss, disabled with values<0")
        self.parser.add_argument("--min_mss_length", "-l", type=int, default=50,
                                 help="Minimal length of maximum scoring segments, ignored if --no_use_mss")
        self.parser.add_argument("--threads", "-t", type=int, default=1, help="Accepted for compatibility (ignored)")
        self.parser.add_argument("--xla", action="store_true", help="Accepted for compatibility (ignored)")
        self.parser.add_argument("-v", "--verbose", action="count", default=0, help="Increase verbosity")
        train = subparsers.add_parser(name="train", formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                      description="Train a deepgrp model (not available in deepgrp_amd)")
        train.add_argument("parameter", type=str)
        train.add_argument("trainfile", type=str)
        train.add_argument("validfile", type=str)
        train.add_argument("bedfile", type=str)
        train.add_argument("--logdir", type=str, default=".")
        train.add_argument("--modelfile", type=str, default="model.hdf5")
        predict = subparsers.add_parser(name="predict", formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                        description="predict using a deepgrp model")
        predict.add_argument("model", type=str, help="Keras model in HDF5 format")
        predict.add_argument("FASTA", nargs="+", type=str, help="Fasta input files ('-' = stdin); a `<fasta>.gz.npz` written by "
                                                                "`preprocess_sequence` is accepted too (one record per file)")
        predict.add_argument("--output", type=str, default="-", help="Output filename")
        predict.add_argument("--no_use_mss", "-m", action="store_true", help="Disable maximum scoring segment algorithm")
        verify = subparsers.add_parser(name="verify", formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                       description="(addition) measure how far the fp16-operand fused kernel is from a plain "
                                                   "fp32 evaluation of the SAME model on the device, on windows of the given "
                                                   "FASTA files (or a random sequence): the accuracy the 1e-3 bound is about")
        verify.add_argument("model", type=str, help="Keras model in HDF5 format")
        verify.add_argument("FASTA", nargs="*", type=str, help="Fasta input files; none = a random ACGT sequence")
        verify.add_argument("--windows", type=int, default=256, help="windows to compare per record (spread evenly)")
        predict.add_argument("--fast", action="store_true",
                             help="(addition) fp16-operand fused kernels: 2-2.5x the default's speed, class probabilities within 1e-3 of "
                                  "fp32 except on ill-conditioned windows (measure with `verify`); the default is fp32-grade (split "
                                  "operands, 1e-5) for every model")
        predict.add_argument("--precise", action="store_true",
                             help="(addition, kept for compatibility) the default: since every model has an fp32-grade fused kernel "
                                  "this flag selects nothing else")
        predict.add_argument("--split_contigs", action="store_true",
                             help="multi-GPU only: spread the windows of EVERY record over all GPUs (for a few huge "
                                  "records) instead of sharding whole records")

    def parse_args(self, argv=None) -> "CommandLineParser":
        argv = list(sys.argv[1:] if argv is None else argv)
        # README form `deepgrp <modelfile> <fastafile>`: insert the sub-command before the first positional
        if not any(a in ("predict", "train", "verify") for a in argv):
            takes_value = {"--batch_size", "-b", "--step_size", "-s", "--xdrop_length", "-x", "--min_mss_length", "-l",
                           "--threads", "-t"}
            i = 0
            while i < len(argv):
                if argv[i] in takes_value:
                    i += 2
                elif argv[i].startswith("-") and argv[i] != "-":
                    i += 1
                else:
                    break
            if i < len(argv):
                argv.insert(i, "predict")
        args = self.parser.parse_args(argv)
        if args.command is None:
            self.parser.error("a sub-command (predict) is required")
        self.threads, self.verbose, self.xla, self.args = args.threads, args.verbose, args.xla, args
        return self

    def setup_tensorflow(self) -> "CommandLineParser":
        """Kept for call-chain compatibility (deepgrp/__main__.py:221-233); nothing to set up."""
        return self

    def set_logging(self) -> "CommandLineParser":
        levels = [logging.WARNING, logging.INFO, logging.DEBUG]
        _LOG.setLevel(levels[min(len(levels) - 1, self.verbose)])
        return self

    def run(self):
        from . import model as dgmodel
        options = dgmodel.Options(min_mss_len=self.args.min_mss_length, batch_size=self.args.batch_size,
                                  xdrop_len=self.args.xdrop_length)
        getattr(self, self.args.command)(self.args, options)

    @staticmethod
    def predict(args: argparse.Namespace, options) -> None:
        """Predict with deepgrp (deepgrp/__main__.py:252-297)."""
        import torch
        import torch.distributed as dist
        from . import model as dgmodel
        from .distributed import gather_records, shard_contigs
        from .fasta import DeviceRecord, read_multi_fasta_device
        from .pipeline import SEGMENT_DTYPE, ContigPipeline, upload_sequence
        from .runner import RecordRunner, rows_text, rows_text_batch

        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        backend = dist.get_backend() if dist.is_initialized() else os.environ.get("DGRP_DIST_BACKEND", "nccl")
        if torch.cuda.is_available():
            local_rank, ndev = int(os.environ.get("LOCAL_RANK", "0")), torch.cuda.device_count()
            if local_rank >= ndev and backend != "gloo":       # (gloo: several ranks may share a GPU -- rehearsals on one card)
                sys.exit(f"rank {rank}: local rank {local_rank} but only {ndev} GPUs visible")
            torch.cuda.set_device(local_rank % max(ndev, 1))
        if world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
            else:
                dist.init_process_group(backend)

        if getattr(args, "precise", False) and getattr(args, "fast", False):
            sys.exit("--precise and --fast exclude each other")
        _LOG.debug("Loading model %s!", args.model)
        model = dgmodel.load_model(args.model, custom_objects={"ReverseComplement": dgmodel.ReverseComplement})
        options.vecsize = model.input_shape[1]
        _LOG.info("Model loading finished successfully!")
        pipe = ContigPipeline(model, args.step_size, options.batch_size, options.min_mss_len, options.xdrop_len,
                              use_mss=not args.no_use_mss, precise=getattr(args, "precise", False),
                              fast=getattr(args, "fast", False))
        _LOG.info("Forward kernel: %s", "plain fp32 kernels (more units than the fused kernels take)" if getattr(model, "fp32_only", False)
                  else "fused, split operands (fp32-grade)" if pipe.split else "fused, fp16 operands")
        outstream = None
        if rank == 0:
            # (headers are carried as bytes through surrogateescape: a Latin-1 header must reach the file as the bytes it was)
            outstream = sys.stdout if args.output == "-" else open(args.output, "w", errors="surrogateescape")

        def records_of(filename):
            if filename.endswith(".npz") and os.path.isfile(filename):
                # (addition) the one-hot `<fasta>.gz.npz` that `preprocess_sequence` writes for training
                # (deepgrp/_scripts/preprocess_sequence.py:71-78), as an alternative input: ONE record, named after the file
                # (the format keeps no header); same stripping of leading/trailing N as one_hot_encode_dna_sequence
                from .preprocessing import load_onehot_npz
                fwd = load_onehot_npz(filename)
                if fwd.size and not (np.isin(fwd, (0, 1)).all() and (fwd.sum(axis=0) == 1).all()):
                    raise ValueError(f"{filename}: `fwd` is not one-hot")
                idx = fwd.argmax(axis=0).astype(np.uint8)
                header = os.path.basename(filename)[:-len(".npz")]
                keep = np.flatnonzero(idx != 4)
                if idx.size == 0:
                    return
                if keep.size == 0:
                    yield header, DeviceRecord(int(idx.size), None, -int(idx.size))        # all N: the reference raises (sequence.pyx:32)
                    return
                st, en = int(keep[0]), int(keep[-1]) + 1
                d_idx = torch.from_numpy(np.ascontiguousarray(idx[st:en])).to(torch.device("cuda", torch.cuda.current_device()))
                yield header, DeviceRecord(st, d_idx, en - st)
                return
            if filename == "-" or not os.path.isfile(filename):
                filestream = sys.stdin if filename == "-" else open(filename, "r")
                try:
                    yield from _read_multi_fasta(filestream)
                finally:
                    if filename != "-":
                        filestream.close()
            else:
                yield from read_multi_fasta_device(filename)

        runner = RecordRunner(pipe)

        try:
            if world == 1:
                import time
                for filename in args.FASTA:
                    _LOG.info("Processing %s", filename)
                    t_file, bases = time.perf_counter(), 0
                    if _LOG.isEnabledFor(logging.DEBUG):
                        # -vv: record by record through the staged form of the same path, a device sync and a clock around every
                        # stage (the reference logs a debug line around each stage of _predict, deepgrp/__main__.py:69-79)
                        for header, rec in records_of(filename):
                            rows, n = CommandLineParser._predict_staged(pipe, header, rec)
                            bases += n
                            outstream.write(rows_text(filename, header, rows))
                    else:
                        for kind, key, rows in runner.results(CommandLineParser._counted(records_of(filename), lambda n: None)):
                            outstream.write(rows_text_batch(filename, key, rows) if kind == "batch" else rows_text(filename, key, rows))
                        bases = CommandLineParser._last_count
                    dt = time.perf_counter() - t_file
                    _LOG.info("%s: %d bases in %.3f s (%.1f Mbp/s; ingest, upload, forward, MSS, segments and TSV text)", filename, bases, dt,
                              bases / max(dt, 1e-9) / 1e6)
            else:
                from .distributed import run_split
                if getattr(args, "split_contigs", False):
                    # every record over all ranks (distributed.run_split): every rank holds every record's class indices and takes
                    # its share of the windows; errors are per record and hit every rank alike
                    records = []
                    for filename in args.FASTA:
                        for header, rec in records_of(filename):
                            records.append((filename, header, rec))
                    parts = []
                    for i, (_f, _h, rec) in enumerate(records):
                        if isinstance(rec, DeviceRecord):
                            if rec.length < 0:
                                raise ValueError("negative dimensions are not allowed")
                            startpos, d_idx = rec.startpos, rec.d_idx
                        else:
                            startpos, d_idx = upload_sequence(rec.encode("utf-8"))
                        parts.append(run_split(pipe, d_idx, startpos, i))
                    allrows = np.concatenate(parts) if parts else np.zeros(0, SEGMENT_DTYPE)
                    if rank == 0:
                        for i, (filename, header, _seq) in enumerate(records):
                            outstream.write(rows_text(filename, header, allrows[allrows["contig"] == i]))
                else:
                    CommandLineParser._predict_sharded(args, runner, records_of, outstream)
                dist.barrier()
        finally:
            # rows already produced reach the file even when a later record raises (the reference leaves that to
            # interpreter shutdown)
            if rank == 0 and args.output != "-":
                outstream.close()

    _last_count = 0

    @staticmethod
    def _counted(records, _cb):
        """Pass (header, record) pairs through, adding up the bases they hold (for the per-file rate of -v)."""
        from .fasta import DeviceRecord
        CommandLineParser._last_count = 0
        for header, rec in records:
            CommandLineParser._last_count += max(rec.length, 0) + max(rec.startpos, 0) if isinstance(rec, DeviceRecord) else len(rec)
            yield header, rec

    @staticmethod
    def _predict_staged(pipe, header, rec):
        """One record through encode -> forward + merge -> scores / MSS / vote (or softmax) -> segments, each stage between device
        syncs, with the reference's debug lines (deepgrp/__main__.py:69-79) carrying the stage's milliseconds.  -> (rows, bases)"""
        import time

        import torch

        from .fasta import DeviceRecord
        from .pipeline import SEGMENT_DTYPE, upload_sequence

        def lap(t):
            torch.cuda.synchronize()
            return (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        _LOG.debug("One hot encoding sequence.")
        if isinstance(rec, DeviceRecord):
            if rec.length < 0:
                raise ValueError("negative dimensions are not allowed")
            startpos, d_idx = rec.startpos, rec.d_idx
        else:
            startpos, d_idx = upload_sequence(rec.encode("utf-8"))
        n = int(d_idx.numel())
        ms_enc = lap(t)
        if n == 0:
            return np.zeros(0, SEGMENT_DTYPE), 0
        t = time.perf_counter()
        _LOG.debug("Start prediction.")
        merged = pipe.merged(d_idx)
        ms_fwd = lap(t)
        _LOG.debug("Finish prediction.")
        t = time.perf_counter()
        if pipe.use_mss:
            _LOG.debug("Applying MSS.")
        labels = pipe.labels(merged)
        ms_post = lap(t)
        t = time.perf_counter()
        rows = pipe.segments(labels, startpos)
        ms_seg = lap(t)
        _LOG.debug("%s: %d bases; encode %.2f ms, forward + merge %.2f ms (%.1f Mbp/s), %s %.2f ms, segments + read-back %.2f ms, %d rows",
                   header, n, ms_enc, ms_fwd, n / max(ms_fwd, 1e-6) / 1e3, "scores + MSS + vote" if pipe.use_mss else "softmax", ms_post,
                   ms_seg, len(rows))
        return rows, n

    @staticmethod
    def _predict_sharded(args, runner, records_of, outstream) -> None:
        """Records sharded over the ranks (the reference's record loop, deepgrp/__main__.py:275-292, carries no state from one
        record to the next).  Ingest is rank-local: the chunk table of every FASTA file comes from host scans of 1/world of its
        bytes per rank, the chunks are shared out longest-first by byte length (runs of short records travel together), and a
        rank reads, uploads and encodes ONLY the byte ranges of its share.  Rank 0 gathers the 24-byte segment records (RCCL)
        and writes them in input order.  Inputs that are not regular FASTA files (stdin, .npz) are parsed by every rank and
        shared out as whole records."""
        import torch
        import torch.distributed as dist

        from . import fasta
        from .distributed import file_chunk_tables, gather_records, plan_file_shares, raise_together
        from .fasta import DeviceRecord
        from .pipeline import SEGMENT_DTYPE
        from .runner import rows_text_batch
        world, rank = dist.get_world_size(), dist.get_rank()
        files = list(args.FASTA)
        sharded = [i for i, f in enumerate(files) if f != "-" and os.path.isfile(f) and not f.endswith(".npz")]
        parsed = []                                        # (file index, record number, header, record), the same on every rank
        for i, f in enumerate(files):
            if i not in sharded:
                parsed += [(i, j, header, rec) for j, (header, rec) in enumerate(records_of(f))]
        length = lambda r: max(r.length, 0) if isinstance(r, DeviceRecord) else len(r)
        tables = file_chunk_tables([files[i] for i in sharded])
        sizes = [os.path.getsize(files[i]) for i in sharded]
        ranges, extras = plan_file_shares(tables, sizes, [length(p[3]) for p in parsed], world)
        uploaded0 = fasta.UPLOAD_STATS["bytes"]

        def my_records():
            work = {}
            for f, a, b in ranges[rank]:
                work.setdefault(sharded[f], []).append((a, b))
            for i in extras[rank]:
                work.setdefault(parsed[i][0], []).append(parsed[i])
            for fi in sorted(work):
                if fi in sharded:
                    for key, header, rec in fasta.ingest_ranges(files[fi], work[fi]):
                        yield ((fi, key), header), rec
                else:
                    for _fi, j, header, rec in work[fi]:
                        yield ((fi, (j, 0)), header), rec

        entries, parts, failure = [], [], None           # entries[local id] = (key, header)
        try:
            for kind, key, rows in runner.results(my_records()):
                if kind == "batch":
                    rows["contig"] += len(entries)
                    entries += key
                else:
                    rows["contig"] = len(entries)
                    entries.append(key)
                parts.append(rows)
        except Exception as e:                  # noqa: BLE001 -- re-raised on every rank together, below
            failure = e
        raise_together(failure)                 # a record that raises (all-N ...) must not leave the others in the gather
        every = [None] * world
        dist.all_gather_object(every, entries)
        order = sorted((key, r, lid) for r, ents in enumerate(every) for lid, (key, _h) in enumerate(ents))
        gid = np.zeros(max(len(entries), 1), np.int32)
        for g, (_key, r, lid) in enumerate(order):
            if r == rank:
                gid[lid] = g
        local = np.concatenate(parts) if parts else np.zeros(0, SEGMENT_DTYPE)
        local["contig"] = gid[local["contig"]]
        allrows = gather_records(local, torch.device("cuda", torch.cuda.current_device()))
        CommandLineParser.last_sharded = {"uploaded_bytes": fasta.UPLOAD_STATS["bytes"] - uploaded0, "records": len(entries),
                                          "file_bytes": int(sum(sizes))}
        _LOG.info("rank %d: %d records, %d of %d file bytes uploaded", rank, len(entries),
                  CommandLineParser.last_sharded["uploaded_bytes"], int(sum(sizes)))
        if rank == 0:
            g = 0
            while g < len(order):                          # one formatter call per input file
                fi, g0 = order[g][0][0], g
                while g < len(order) and order[g][0][0] == fi:
                    g += 1
                lo, hi = np.searchsorted(allrows["contig"], [g0, g])
                rows = allrows[lo:hi].copy()
                rows["contig"] -= g0
                outstream.write(rows_text_batch(files[fi], [every[r][lid][1] for _k, r, lid in order[g0:g]], rows))

    last_sharded: dict = {}

    @staticmethod
    def verify(args: argparse.Namespace, options) -> None:
        """One line per record and fused kernel ("split" = the default where it exists, "fp16" = --fast): largest
        |p_fused - p_fp32| over the compared windows, and how many per-base argmax calls differ.  Exit status 1 if the
        default kernel exceeds 1e-3 on any record."""
        from . import model as dgmodel
        from .fasta import DeviceRecord, read_multi_fasta_device
        from .pipeline import upload_sequence
        model = dgmodel.load_model(args.model, custom_objects={"ReverseComplement": dgmodel.ReverseComplement})
        levels = ([("split", 1)] if model.supports_split else []) + [("fp16", 0)]
        results = []

        def measure(filename, header, d_idx):
            for name, level in levels:
                results.append((filename, header, name, model.check_accuracy(d_idx, args.step_size, args.windows, level=level)))

        if not args.FASTA:
            measure("<random>", "ACGT", None)
        for filename in args.FASTA:
            if filename == "-" or not os.path.isfile(filename):
                stream = sys.stdin if filename == "-" else open(filename, "r")
                records = list(_read_multi_fasta(stream))
            else:
                records = list(read_multi_fasta_device(filename))
            for header, rec in records:
                d_idx = rec.d_idx if isinstance(rec, DeviceRecord) else upload_sequence(rec.encode("utf-8"))[1]
                if d_idx.numel() <= model.input_shape[1]:
                    continue                                                # no window (prediction.py:31)
                measure(filename, header, d_idx)
        bad = False
        for filename, header, kernel, r in results:
            sys.stdout.write(f"{filename}\t{header}\t{kernel}\t{r['max_abs_diff']:.3e}\t{r['windows_checked']}\t{r['argmax_flips']}\t"
                             f"{'ok' if r['within_1e-3'] else 'ABOVE 1e-3'}\n")
            # the verdict is about the kernel `predict` uses by default for this model
            bad = bad or (kernel == levels[0][0] and not r["within_1e-3"])
        if bad:
            sys.exit(1)

    @staticmethod
    def train(args: argparse.Namespace, options) -> None:
        sys.exit("deepgrp_amd implements the prediction path only; train with the reference (TensorFlow) and "
                 "pass the saved .hdf5 to `predict`")


def main(argv=None):
    CommandLineParser().parse_args(argv).set_logging().setup_tensorflow().run()


if __name__ == "__main__":
    main()
