#!/usr/bin/env python3
"""Generate deepgrp_amd/csrc/gru_split2_phase.inc: the instruction order of one phase of gru_split2_kernel.

A phase = the 82 MFMAs of tile X's step (4 input-projection + 8 k-steps x 9 + 6 Dense) with tile Y's epilogue cut into
single operations that sit in the gaps between them (gru_split2.hip explains why).  The macros are defined there:

    M_IN(i) M_K(k, j) M_D(i)   X's MFMAs (asm volatile: stay in program order)
    PF(k) RDD                  X's LDS reads: fragments of k-step k, Dense operands
    XP(op)                     Y: one-hot operand of its next step (0: read the base, 1: build the operand)
    DS(i)                      Y: store Dense partial i of the previous step
    G(e, op)                   Y: link `op` (0..11) of element e's gate chain (split_gate_op)
    PB(g, op)                  Y: publish group g (elements 4g..4g+3): 0 state -> h, 1 hi = fp16(h), 2 h - hi, 3 lo = fp16(h - hi), 4 two LDS stores
    BAR                        s_waitcnt lgkmcnt(0) + s_barrier, flip Y's ping-pong
    RD0                        Y: first fragments of its next step
    FN(op)                     Y: softmax + max-merge of the previous step's logits (0..12)
    GAP                        sched_barrier(0)

Cost model (cycles of the SIMD's issue port, MI355X_MICROARCH.md "vector-instruction ISSUE cost"): transcendental 8, plain
VALU 4; an MFMA 32x32x16 occupies the pipe for 32 cycles and the port for 8, so up to BUDGET = 24 cycles of other work per
gap are hidden; a 16x16x32 gap hides 8.  Operations are taken from two queues (before / after the barrier) in order and
packed greedily; nothing that reads the previous phase's accumulators goes into the first FREE_HEAD gaps (MFMA result ->
VALU read needs wait states the compiler does not pad behind inline asm).

    python tools/gen_split2_schedule.py [--skew 3] [--bar-k 6] [--budget 24] [--report]
"""
import argparse
import os

T, P = 8, 4
GATE_T = {0, 1, 3, 5, 10}            # transcendental links of the ONERCP chain (the other form has one more: link 8)


def gate_pipeline(skew):
    """Gate chains of the 16 elements, element e lagging e*skew links behind element 0 (independent chains interleave);
    a group of four elements is published as soon as its last chain is done."""
    ops, done = [], set()
    tau = 0
    while len(done) < 16:
        for e in range(16):
            op = tau - skew * e
            if 0 <= op < 12:
                ops.append((f"G({e}, {op})", T if op in GATE_T else P))
                if op == 11:
                    done.add(e)
                    if e % 4 == 3:
                        g = e // 4
                        ops += [(f"PB({g}, 0)", 16), (f"PB({g}, 1)", 8), (f"PB({g}, 2)", 16), (f"PB({g}, 3)", 8), (f"PB({g}, 4)", 12)]
        tau += 1
    return ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skew", type=int, default=3)
    ap.add_argument("--bar-k", type=int, default=6, help="the barrier follows the last MFMA of this k-step")
    ap.add_argument("--budget", type=int, default=24)
    ap.add_argument("--dense-budget", type=int, default=8)
    ap.add_argument("--free-head", type=int, default=2)
    ap.add_argument("--report", action="store_true")
    ap.add_argument("-o", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deepgrp_amd", "csrc",
                                               "gru_split2_phase.inc"))
    a = ap.parse_args()

    slots = [(f"M_IN({i})", a.budget, ("in", i)) for i in range(4)]
    for k in range(8):
        slots += [(f"M_K({k}, {j})", a.budget, ("k", k, j)) for j in range(9)]
    slots += [(f"M_D({i})", a.dense_budget, ("d", i)) for i in range(6)]
    bar_slot = 4 + 9 * (a.bar_k + 1) - 1                           # index of the slot the barrier closes

    pre = [(f"DS({i})", 6) for i in range(4)] + gate_pipeline(a.skew)
    post = [("RD0", 8), ("RDD", 16), ("FN(0)", 16), ("FN(1)", 16)] + [(f"FN({i})", 8) for i in range(2, 6)] + [("FN(6)", 16)] + \
           [(f"FN({i})", 8) for i in range(7, 11)] + [("FN(11)", 12), ("FN(12)", 16)]
    pinned = {0: [("XP(0)", 4)], 1: [("XP(1)", 40)]}
    for k in range(7):                                             # fragments of k-step k+1: early in k-step k
        pinned.setdefault(4 + 9 * k + 1, []).append((f"PF({k + 1})", 8))

    lines, rep = [], []
    over = 0
    for si, (mf, budget, _) in enumerate(slots):
        items, used = [], 0
        for name, c in pinned.get(si, []):
            items.append(name); used += c
        queue = pre if si <= bar_slot else post
        if si >= a.free_head:
            while queue and (used + queue[0][1] <= budget or used == 0 or (si == bar_slot and queue is pre)):
                name, c = queue.pop(0)
                items.append(name); used += c
        if si == bar_slot:
            items.append("BAR")
        over += max(0, used - budget)
        lines.append(f"{mf} " + " ".join(items) + (" " if items else "") + "GAP")
        rep.append((mf, used, budget, items))
    if pre or post:
        # whatever is left goes behind the last MFMA (exposed)
        rest = [n for n, _ in pre + post]
        over += sum(c for _, c in pre + post)
        lines.append(" ".join(rest) + " GAP")
    hdr = ["// generated by tools/gen_split2_schedule.py " + " ".join(f"--{k.replace('_', '-')} {v}" for k, v in sorted(vars(a).items())
                                                                     if k not in ("report", "o")),
           f"// {len(slots)} MFMA gaps; issue-port cycles beyond the gaps' budgets: {over}"]
    with open(a.o, "w") as fh:
        fh.write("\n".join(hdr + lines) + "\n")
    if a.report:
        for mf, used, budget, items in rep:
            print(f"{mf:12s} {used:3d}/{budget:2d}  {' '.join(items)}")
    print(f"wrote {os.path.relpath(a.o)}: over-budget cycles {over}, left over: {len(pre) + len(post)} ops")


if __name__ == "__main__":
    main()
