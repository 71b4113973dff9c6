#!/usr/bin/env python3
"""Fit the DeepGRP architecture (deepgrp/model.py:293-336, no attention) to the synthetic
chromosomes' planted tandem repeats, on CPU with torch, and store the tensors in Keras layout
(deepgrp_amd/data/synthetic_gru128.npz).  Benchmark tooling only: it gives bench.py a model whose
output looks like a genome annotation (long confident background, confident repeat runs) instead
of the stationary noise random weights produce.  Training is NOT part of the product."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepgrp_amd import synthetic  # noqa: E402

U, T, C = 128, 200, 5
torch.manual_seed(0)
torch.set_num_threads(8)
COMP = torch.tensor([3, 2, 1, 0, 4])


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.gru = torch.nn.GRU(5, U, batch_first=True)
        self.ff = torch.nn.Linear(U, C)

    def forward(self, idx):                       # idx [B, T] int64
        x = torch.nn.functional.one_hot(idx, 5).float()
        xr = torch.nn.functional.one_hot(COMP[idx.flip(1)], 5).float()
        f, _ = self.gru(x)
        r, _ = self.gru(xr)
        return self.ff((f + r) / 2)               # logits [B, T, C]


def batches(rng, idx, lab, batch):
    n = idx.size
    rep = np.flatnonzero(lab > 0)
    while True:
        starts = np.where(rng.random(batch) < 0.6, rng.choice(rep, batch) - rng.integers(0, T, batch),
                          rng.integers(0, n - T, batch))
        starts = np.clip(starts, 0, n - T - 1)
        sel = starts[:, None] + np.arange(T)[None, :]
        yield torch.from_numpy(idx[sel].astype(np.int64)), torch.from_numpy(lab[sel].astype(np.int64))


def main(steps=int(os.environ.get("STEPS", 700))):
    idx, lab = synthetic.synthetic_truth(3_000_000, contig=100, flank=0)
    rng = np.random.default_rng(1)
    net = Net()
    opt = torch.optim.Adam(net.parameters(), lr=3e-3)
    wts = torch.tensor([0.5, 1.0, 1.0, 1.0, 1.0])
    t0 = time.time()
    for step, (x, y) in zip(range(steps), batches(rng, idx, lab, 96)):
        logits = net(x)
        loss = torch.nn.functional.cross_entropy(logits.reshape(-1, C), y.reshape(-1), weight=wts)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        if step % 25 == 0:
            acc = (logits.argmax(-1) == y).float().mean().item()
            print(f"step {step} loss {loss.item():.4f} acc {acc:.3f} {time.time()-t0:.0f}s", flush=True)
        if step == int(steps * 0.7):
            for g in opt.param_groups:
                g["lr"] = 1e-3
    # torch (r|z|n rows) -> Keras (z|r|h columns)
    perm = np.concatenate([np.arange(U, 2 * U), np.arange(0, U), np.arange(2 * U, 3 * U)])
    g = net.gru
    out = dict(kernel=g.weight_ih_l0.detach().numpy()[perm].T.copy(),
               recurrent_kernel=g.weight_hh_l0.detach().numpy()[perm].T.copy(),
               bias=np.stack([g.bias_ih_l0.detach().numpy()[perm], g.bias_hh_l0.detach().numpy()[perm]]),
               ff_kernel=net.ff.weight.detach().numpy().T.copy(), ff_bias=net.ff.bias.detach().numpy().copy())
    out = {k: np.ascontiguousarray(v, np.float32) for k, v in out.items()}
    path = os.path.join(ROOT, "deepgrp_amd", "data", "synthetic_gru128.npz")
    np.savez_compressed(path, **out)
    print("saved", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
