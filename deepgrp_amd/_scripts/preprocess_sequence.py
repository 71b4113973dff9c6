#!/usr/bin/env python3
"""gzip FASTA -> `<fasta>.npz` with the one-hot int8 [5, N] array `fwd` and the md5 of the sequence lines in `hash`.
Mirror of the reference's `preprocess_sequence` console script (deepgrp/_scripts/preprocess_sequence.py; SURVEY 8f
N4); the output is what deepgrp_amd.preprocessing.load_onehot_npz reads.

Kept behaviour: single-record files (a later header only replaces the name, the sequences run together); lines are
stripped, the md5 runs over the stripped sequence lines as they are in the file while the sequence itself is
upper-cased; rows A, C, G, T, N -- any other letter is a KeyError; the file is rewritten only when the stored hash
differs or with --force."""
import argparse
import gzip
import hashlib
import sys
from typing import BinaryIO, List, Optional, Tuple

import numpy as np

ROW_OF = {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}


def fastaparser(stream: BinaryIO) -> Tuple[str, str, str]:
    """(header, md5 hex digest, upper-cased sequence) of a FASTA byte stream."""
    parts: List[str] = []
    digest = hashlib.md5()
    header = None
    for raw in stream:
        line = raw.strip()
        if line[0] == ord(">"):                     # an empty line is an IndexError, as upstream
            header = line[1:].decode()
        else:
            parts.append(line.decode().upper())
            digest.update(line)
    if header is None:
        raise UnboundLocalError("local variable 'header' referenced before assignment")
    return header, digest.hexdigest(), "".join(parts)


def one_hot(seq: str) -> np.ndarray:
    """int8 [5, len(seq)] with a single 1 per column; KeyError on letters outside ACGTN."""
    codes = np.frombuffer(seq.encode("latin-1", "replace"), np.uint8)
    lut = np.full(256, 255, np.uint8)
    for letter, row in ROW_OF.items():
        lut[ord(letter)] = row
    rows = lut[codes]
    bad = np.flatnonzero(rows == 255)
    if bad.size:
        raise KeyError(seq[int(bad[0])])
    enc = np.zeros((len(ROW_OF), len(seq)), dtype=np.int8)
    enc[rows, np.arange(len(seq))] = 1
    return enc


def main(argv: Optional[List[str]] = None) -> None:
    ap = argparse.ArgumentParser(description="Format fasta file to onehot encoded sequences")
    ap.add_argument("FASTAFILE", type=str, help="Fastafile (gzip)")
    ap.add_argument("--force", action="store_true", help="forces recreation even if files not changed")
    args = ap.parse_args(argv)
    try:
        with gzip.open(args.FASTAFILE, "rb") as fh:
            _, digest, seq = fastaparser(fh)
    except IOError:
        sys.stderr.write("Could not open file!\n")
        sys.exit(1)
    rebuild = args.force
    try:
        with np.load(args.FASTAFILE + ".npz", allow_pickle=False) as old:
            if digest != old["hash"][0]:
                rebuild = True
    except (IOError, KeyError):
        rebuild = True
    if rebuild:
        np.savez_compressed(args.FASTAFILE, fwd=one_hot(seq), hash=np.array([digest]))


if __name__ == "__main__":
    main()
