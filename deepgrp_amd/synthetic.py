"""Synthetic inputs of BASELINE.json's shape: there are no trained
weights or genomes offline, so benchmarks, the smoke test and parity tests use
seeded random chromosomes and Keras-initialiser weights."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def synthetic_chromosome(n_bases: int, contig: int = 0, n_frac: float = 0.01, flank: int = 10_000,
                         structured: bool = True) -> bytes:
    """Upper-case sequence bytes: leading/trailing N blocks of `flank` (exercises startpos),
    N injected at `n_frac`, bases i.i.d. uniform; with `structured`, stretches of
    low-complexity repeats (random short motifs tandemly repeated) are planted so that the
    model's output is not stationary noise.  numpy.random.default_rng(seed=20240+contig)."""
    rng = np.random.default_rng(20240 + contig)
    body = n_bases - 2 * flank
    if body <= 0:
        flank, body = 0, n_bases
    seq = rng.integers(0, 4, size=body, dtype=np.uint8)
    if structured and body > 4000:
        pos = 0
        while pos < body:
            pos += int(rng.integers(2_000, 40_000))
            ln = int(rng.integers(300, 6_000))
            if pos + ln >= body:
                break
            motif = rng.integers(0, 4, size=int(rng.integers(1, 7)), dtype=np.uint8)
            seq[pos:pos + ln] = np.resize(motif, ln)
            pos += ln
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    if n_frac > 0:
        seq = seq.copy()
        seq[rng.random(body) < n_frac] = 4
    out = np.concatenate([np.full(flank, ord("N"), np.uint8), lut[seq], np.full(flank, ord("N"), np.uint8)])
    return out.tobytes()


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
