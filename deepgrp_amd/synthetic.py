"""Synthetic inputs of BASELINE.json's shape: there are no trained
weights or genomes offline, so benchmarks, the smoke test and parity tests use
seeded random chromosomes and Keras-initialiser weights."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def _planted(n_bases: int, contig: int, n_frac: float, flank: int, structured: bool):
    """(class index per base incl. flanks as 4, truth label per base)."""
    rng = np.random.default_rng(20240 + contig)
    body = n_bases - 2 * flank
    if body <= 0:
        flank, body = 0, n_bases
    seq = rng.integers(0, 4, size=body, dtype=np.uint8)
    truth = np.zeros(body, np.uint8)
    if structured and body > 4000:
        pos = 0
        while pos < body:
            pos += int(rng.integers(500, 12_000))
            ln = int(rng.integers(150, 4_000))
            if pos + ln >= body:
                break
            period = int(rng.integers(1, 7))
            motif = rng.integers(0, 4, size=period, dtype=np.uint8)
            if period > 1 and (motif == motif[0]).all():
                motif[-1] = (motif[0] + 1) % 4
            unit = np.resize(motif, ln)
            mut = rng.random(ln) < 0.03                          # a few substitutions, like real tandem repeats
            unit = np.where(mut, rng.integers(0, 4, size=ln, dtype=np.uint8), unit)
            seq[pos:pos + ln] = unit
            truth[pos:pos + ln] = PERIOD_CLASS[period]
            pos += ln
    if n_frac > 0:
        seq = seq.copy()
        seq[rng.random(body) < n_frac] = 4
    idx = np.concatenate([np.full(flank, 4, np.uint8), seq, np.full(flank, 4, np.uint8)])
    lab = np.concatenate([np.zeros(flank, np.uint8), truth, np.zeros(flank, np.uint8)])
    return idx, lab


# repeat class of a planted tandem repeat by its period (4 repeat classes like repeats_to_search)
PERIOD_CLASS = {1: 1, 2: 1, 3: 2, 4: 3, 5: 3, 6: 4}


def synthetic_chromosome(n_bases: int, contig: int = 0, n_frac: float = 0.01, flank: int = 10_000,
                         structured: bool = True) -> bytes:
    """Upper-case sequence bytes: leading/trailing N blocks of `flank` (exercises startpos),
    N injected at `n_frac`, bases i.i.d. uniform; with `structured`, tandem repeats of period
    1-6 (150-4000 bp, 3 % substitutions) are planted every 0.5-12 kb -- the "repeats" the
    trained synthetic model (data/synthetic_gru128.npz) calls.
    numpy.random.default_rng(seed=20240+contig)."""
    idx, _ = _planted(n_bases, contig, n_frac, flank, structured)
    return np.frombuffer(b"ACGTN", dtype=np.uint8)[idx].tobytes()


def synthetic_truth(n_bases: int, contig: int = 0, n_frac: float = 0.01, flank: int = 10_000):
    """(class index uint8 [n], truth label uint8 [n]) of the same chromosome."""
    return _planted(n_bases, contig, n_frac, flank, True)


def trained_weights(path: Optional[str] = None) -> Dict[str, Optional[np.ndarray]]:
    """The small model tools/train_synth_model.py fitted (torch, CPU) to call the planted tandem
    repeats: u=128, T=200, 5 classes, no attention -- genome-like output (confident background,
    confident repeat runs) for benchmarks; there is no trained DeepGRP model offline."""
    import os
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "synthetic_gru128.npz")
    z = np.load(path, allow_pickle=False)
    return dict(kernel=z["kernel"], recurrent_kernel=z["recurrent_kernel"], bias=z["bias"], ff_kernel=z["ff_kernel"],
                ff_bias=z["ff_bias"], scale=None)


def synthetic_weights(units: int = 128, classes: int = 5, attention: bool = False, seed: int = 7,
                      gain: float = 1.0) -> Dict[str, Optional[np.ndarray]]:
    """Tensors in Keras layout with the initialisers recorded in the reference's
    tests/test_model.json: glorot_uniform kernel / FF, orthogonal recurrent kernel, zero-mean
    small biases, glorot attention scale; `gain` scales the matrices (3.0 = the "structured"
    set giving long confident runs)."""
    rng = np.random.default_rng(seed)
    u = units

    def glorot(shape):
        lim = np.sqrt(6.0 / (shape[0] + shape[-1]))
        return rng.uniform(-lim, lim, size=shape)

    kernel = glorot((5, 3 * u)) * gain
    q, r = np.linalg.qr(rng.normal(size=(3 * u, u)))
    recurrent = (q * np.sign(np.diag(r))).T.copy() * gain
    bias = rng.normal(scale=0.05, size=(2, 3 * u))
    ffk = glorot(((2 if attention else 1) * u, classes)) * gain
    ffb = rng.normal(scale=0.05, size=(classes,))
    scale = glorot((u, 1))[:, 0] if attention else None
    f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    return dict(kernel=f32(kernel), recurrent_kernel=f32(recurrent), bias=f32(bias), ff_kernel=f32(ffk),
                ff_bias=f32(ffb), scale=f32(scale))
