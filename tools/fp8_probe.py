"""VERDICT r02 item 5, the accuracy half: what would the headline kernel's class probabilities be if its two low-order passes
(U_hi.h_lo and U_lo.h_hi) ran on block-scaled fp8 MFMAs (v_mfma_scale_f32_16x16x128_f8f6f4, e4m3 operands) instead of fp16?

The recurrence of the benchmark model is evaluated in torch on the GPU with the operand roundings of each candidate contraction and
float64 accumulation -- an EMULATION of the arithmetic, not a kernel: it isolates what the operand formats cost.  Validated by its
first rows, which re-create the measured table of tools/twopass_probe.py (three fp16 passes: median 3.6e-7, max 1.9e-5; a pass
dropped: median 5e-5 .. 8e-5, max 8e-3 .. 1e-2 -- DESIGN.md 1).  Same windows as that probe and as bench.py's `accuracy` block: 4096
windows spread over the 50 Mbp synthetic chromosome, the benchmark's fitted model, yardstick = the same recurrence in float64.

    python tools/fp8_probe.py [Mbp] [windows]
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepgrp_amd import synthetic                                   # noqa: E402
from deepgrp_amd.pipeline import upload_sequence                    # noqa: E402

mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 50
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T, S, U, C = 200, 50, 128, 5
dev = torch.device("cuda", 0)
w = synthetic.trained_weights()
st, d_idx = upload_sequence(synthetic.synthetic_chromosome(int(mbp * 1e6)))
total = len(range(0, d_idx.numel() - T, S))
chunk = 64
starts = sorted({int(x) for x in np.linspace(0, total - chunk, windows // chunk)})
wins = torch.tensor([s0 + k for s0 in starts for k in range(chunk)], device=dev)
pos = wins[:, None] * S + torch.arange(T, device=dev)[None, :]
base_f = d_idx[pos].long()                                            # [W, T]
comp = torch.tensor([3, 2, 1, 0, 4], device=dev)
base_r = comp[base_f.flip(1)]                                         # reverse complement (model.py:266-279)
bases = torch.cat([base_f, base_r], 0)                                # [2W, T]
f64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64, device=dev)
Wx, Uk, bias, Wd, bd = f64(w["kernel"]), f64(w["recurrent_kernel"]), f64(w["bias"]), f64(w["ff_kernel"]), f64(w["ff_bias"])
# the kernels fold the exp2 scale into the weights before they split them: z, r columns -log2 e, candidate 2 log2 e
gs = torch.cat([torch.full((2 * U,), -1.4426950408889634), torch.full((U,), 2.8853900817779268)]).double().to(dev)
Us = Uk * gs[None, :]


def fp16(x):
    return x.to(torch.float16).to(torch.float64)


def block_fp8(x, kdim, dtype, top):
    """x with an e8m0 (power of two) scale per block of 32 elements along its contraction dimension `kdim` and fp8 elements of `dtype`
    (OCP MX block scaling as v_mfma_scale_*_f8f6f4 takes it; the scale puts the block's largest magnitude below `top`)."""
    y = x.movedim(kdim, -1)
    shp = y.shape
    y = y.reshape(*shp[:-1], shp[-1] // 32, 32)
    amax = y.abs().amax(dim=-1, keepdim=True).clamp_min(1e-300)
    scale = torch.exp2(torch.ceil(torch.log2(amax / top)))
    q = (y / scale).to(torch.float32).to(dtype).to(torch.float64) * scale
    return q.reshape(shp).movedim(-1, kdim)


def e4m3(x, kdim):
    return block_fp8(x, kdim, torch.float8_e4m3fn, 448.0)


def e5m2(x, kdim):
    return block_fp8(x, kdim, torch.float8_e5m2, 57344.0)


U_hi = fp16(Us)
U_lo = fp16(Us - U_hi)


def contraction(kind):
    """h [2W, U] float64 (the fp32 state, widened) -> h.U' [2W, 3U] under the operand formats of `kind`."""
    if kind == "exact":
        return lambda h: h @ Us
    def f(h):
        h32 = h.to(torch.float32).to(torch.float64)
        h_hi = fp16(h32)
        h_lo = fp16(h32 - h_hi)
        out = h_hi @ U_hi
        if kind == "three fp16 passes (shipped)":
            return out + h_lo @ U_hi + h_hi @ U_lo
        if kind == "without U_lo.h_hi":
            return out + h_lo @ U_hi
        if kind == "without U_hi.h_lo":
            return out + h_hi @ U_lo
        if kind == "one fp16 pass (--fast)":
            return out
        q = e4m3 if "e4m3" in kind else e5m2
        if "fp16 U_lo pass" in kind:                                  # only U_hi.h_lo on fp8
            return out + q(h_lo, 1) @ q(U_hi, 0) + h_hi @ U_lo
        if "fp16 h_lo pass" in kind:                                  # only U_lo.h_hi on fp8
            return out + h_lo @ U_hi + q(h_hi, 1) @ q(U_lo, 0)
        return out + q(h_lo, 1) @ q(U_hi, 0) + q(h_hi, 1) @ q(U_lo, 0)
    return f


def forward(kind):
    mm = contraction(kind)
    xp = (Wx + bias[0][None, :]) * gs[None, :]                       # input projection rows, exp2 domain
    br = bias[1] * gs
    h = torch.zeros((bases.shape[0], U), dtype=torch.float64, device=dev)
    hs = []
    for t in range(T):
        x = xp[bases[:, t]]
        g = mm(h) + br[None, :]
        z = 1.0 / (1.0 + torch.exp2(x[:, :U] + g[:, :U]))
        r = 1.0 / (1.0 + torch.exp2(x[:, U:2 * U] + g[:, U:2 * U]))
        hh = 1.0 - 2.0 / (1.0 + torch.exp2(x[:, 2 * U:] + r * g[:, 2 * U:]))
        h = z * h + (1.0 - z) * hh
        if kind != "exact":
            h = h.to(torch.float32).to(torch.float64)                 # the kernels keep the state in fp32
        hs.append(h)
    hs = torch.stack(hs, 1)                                           # [2W, T, U]
    W_ = hs.shape[0] // 2
    avg = 0.5 * (hs[:W_] + hs[W_:])                                   # same step index in both passes
    return torch.softmax(avg @ Wd + bd, dim=2)


ref = forward("exact")
kinds = ["three fp16 passes (shipped)", "without U_lo.h_hi", "without U_hi.h_lo", "one fp16 pass (--fast)",
         "both low passes e4m3", "both low passes e5m2", "U_hi.h_lo e4m3, fp16 U_lo pass", "U_lo.h_hi e4m3, fp16 h_lo pass"]
print(f"# {wins.numel()} windows of a {mbp:g} Mbp synthetic chromosome, benchmark model; |p - p_float64| per window = max over steps and classes")
print(f"{'recurrent contraction':42s} {'median':>10s} {'q99':>10s} {'max':>10s} {'windows > 2e-4':>15s} {'> 1e-3':>8s} {'argmax flips':>13s}")
for kind in kinds:
    p = forward(kind)
    d = (p - ref).abs().amax(dim=(1, 2))
    flips = int((p.argmax(2) != ref.argmax(2)).sum())
    q = torch.quantile(d, torch.tensor([0.5, 0.99], dtype=torch.float64, device=dev)).cpu().numpy()
    print(f"{kind:42s} {q[0]:10.2e} {q[1]:10.2e} {float(d.max()):10.2e} {int((d > 2e-4).sum()):15d} {int((d > 1e-3).sum()):8d} {flips:13d}", flush=True)
