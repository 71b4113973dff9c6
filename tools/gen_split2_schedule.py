#!/usr/bin/env python3
"""Generate deepgrp_amd/csrc/gru_split2_phase.inc: the instruction order of one phase of gru_split2_kernel.

A phase = the 150 MFMAs (v_mfma_f32_16x16x32_f16) of tile X's step -- 4 k-steps of 32 x 36 (3 passes x 3 gates x 2 unit
halves x 2 row halves) + 6 Dense -- with tile Y's epilogue cut into single operations that sit in the gaps between them
(gru_split2.hip explains why).  The macros are defined there:

    M_K(ks, n) M_D(i)          X's MFMAs (asm volatile: stay in program order)
    PF(ks) RDD                 X's LDS reads: fragments of k-step ks, Dense operands
    AXL(sub)                   Y: the candidate's input projection (table row) of sub-tile sub
    DS(i)                      Y: store Dense partial i of the previous step
    G(e, op)                   Y: link `op` (0..11) of element e's gate chain (split_gate_op)
    PB(g, op)                  Y: publish sub-tile g (elements 4g..4g+3): 0 state -> h, 1 hi = fp16(h), 2 h - hi, 3 lo = fp16(h - hi), 4 two LDS stores
    XP(op)                     Y: bases of its next step (0: read the two bytes, 1: table row offsets)
    BAR                        s_waitcnt lgkmcnt(0) + s_barrier, flip Y's ping-pong
    RD0 CI(g, sub)             Y's next step: first fragments; accumulator sub of gate g starts as its table row
    FN(op)                     X: softmax + max-merge of ITS logits of two steps ago (0..12; stored a phase and a barrier earlier)
    GAP                        sched_barrier(0)
    ST(i)                      diagnostic build only (-DDGRP_STAMP): add the cycles since the last stamp to section i

Cost model (cycles of the SIMD's issue port, MI355X_MICROARCH.md "vector-instruction ISSUE cost"): transcendental 8, plain
VALU 4, LDS instruction 4.  A 16x16x32 MFMA occupies the pipe for 16 cycles and the port for 8; the epilogue's ~1500 port
cycles do not fit the 150 x 8 free ones, so the stream is paced by the port and the aim is an even spread: BUDGET cycles of
work per gap, taken in order from the queue in front of the barrier and the one behind it (layout 2: only RD0 and FN's last,
branching link are behind it).  Nothing that reads the previous
phase's accumulators or Dense result goes into the first FREE_HEAD gaps (MFMA result -> VALU read needs wait states the
compiler does not pad behind inline asm).

    python tools/gen_split2_schedule.py [--skew 3] [--bar 104] [--budget 12] [--report]
"""
import argparse
import os

T, P = 8, 4
GATE_T = {0, 1, 3, 5, 10}            # transcendental links of the ONERCP chain (the other form has one more: link 8)


def gate_pipeline(skew, early_ci=False):
    """Gate chains of the 16 elements, element e lagging e*skew links behind element 0 (independent chains interleave);
    a sub-tile of four elements is published as soon as its last chain is done.  early_ci: the accumulators of a sub-tile
    restart from their table rows (CI) as soon as its four chains have read them (link 4 is the last reader)."""
    ops, done = [], set()
    tau = 0
    while len(done) < 16:
        for e in range(16):
            op = tau - skew * e
            if 0 <= op < 12:
                ops.append((f"G({e}, {op})", T if op in GATE_T else P))
                if early_ci and op == 4 and e % 4 == 3:
                    ops += [(f"CI({g}, {e // 4})", 4) for g in range(3)]
                if op == 11:
                    done.add(e)
                    if e % 4 == 3:
                        g = e // 4
                        ops += [(f"PB({g}, 0)", 16), (f"PB({g}, 1)", 8), (f"PB({g}, 2)", 16), (f"PB({g}, 3)", 8), (f"PB({g}, 4)", 8)]
        tau += 1
    return ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skew", type=int, default=3)
    ap.add_argument("--late-ci", action="store_true", help="layout 2 with CI behind the barrier (bisecting)")
    ap.add_argument("--late-fn", action="store_true", help="layout 2 with FN behind the barrier (bisecting)")
    ap.add_argument("--early-emit", action="store_true",
                    help="layout 2 with FN's last link (the merge / store, the one link with branches in it) in front of the barrier "
                         "as well: with nothing but RD0 behind the barrier the register allocator split the accumulators' live ranges "
                         "and put v_mov copies directly in front of MFMAs (tools/lint_split2_isa.py hazard A, wrong results)")
    ap.add_argument("--late-ci-sub", type=int, default=-1, help="bisecting: CI of sub-tiles >= this behind the barrier")
    ap.add_argument("--bar", type=int, default=140, help="the barrier follows this many of the 144 recurrent MFMAs")
    ap.add_argument("--layout", type=int, default=2,
                    help="1: softmax/merge (FN) and the accumulators' restart (CI) behind the barrier; 2: FN spread over the whole "
                         "phase (it finishes the MFMA tile's logits of two steps ago: nothing in this phase feeds it) and CI as soon "
                         "as a sub-tile's chains have read their accumulators, so that only RD0 waits for the barrier")
    ap.add_argument("--budget", type=float, default=0, help="port cycles per gap in front of the barrier (0 = the queue's average)")
    ap.add_argument("--post-budget", type=float, default=0)
    ap.add_argument("--free-head", type=int, default=4)
    ap.add_argument("--xp1", type=int, default=12, help="gap that turns Y's next bases into table offsets (the bytes are read in gap 0)")
    ap.add_argument("--report", action="store_true")
    ap.add_argument("-o", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deepgrp_amd", "csrc",
                                               "gru_split2_phase.inc"))
    a = ap.parse_args()

    slots = []                                                     # (macro, key)
    for ks in range(4):
        for n in range(36):
            slots.append((f"M_K({ks}, {n})", ("k", ks, n)))
            if n % 12 == 11 and 3 * ks + n // 12 in (1, 3, 5, 7, 9, 10):      # one link of the Dense chain every other pass
                slots.append((f"M_D({[1, 3, 5, 7, 9, 10].index(3 * ks + n // 12)})", ("d",)))
    nrec = 0
    bar_slot = None
    for i, (_, key) in enumerate(slots):
        if key[0] == "k":
            nrec += 1
            if nrec == a.bar:
                bar_slot = i
    # queue in front of the barrier: Dense partials out, the gate pipeline; behind it: Y's next step, then softmax/merge
    fn = [("FN(0)", 8), ("FN(1)", 16)] + [(f"FN({i})", 8) for i in range(2, 6)] + [("FN(6)", 16)] + \
         [(f"FN({i})", 8) for i in range(7, 11)] + [("FN(11)", 12), ("FN(12)", 16)]
    early_ci = a.layout == 2 and not a.late_ci
    early_fn = a.layout == 2 and not a.late_fn
    pre = [(f"DS({i})", 4) for i in range(4)] + gate_pipeline(a.skew, early_ci=early_ci)
    post = [("RD0", 16)]
    head = []
    if not early_ci:
        post += [(f"CI({g}, {sub})", 4) for g in range(3) for sub in range(4)]
    if a.late_ci_sub >= 0:
        late = [x for x in pre if x[0].startswith("CI") and int(x[0][-2]) >= a.late_ci_sub]
        pre = [x for x in pre if x not in late]
        post += late
    if early_fn and not a.early_emit:
        post += fn[-1:]
        fn = fn[:-1]
    if early_fn:
        # FN(0) (an LDS read) goes into the head gaps, which take nothing that touches the previous phase's accumulators; the
        # other links at even distances through the queue
        head = fn[:1]
        rest = fn[1:]
        for i, item in enumerate(rest):
            pre.insert(int((i + 0.5) * len(pre) / len(rest)), item)
    else:
        post += fn[:1] + [("-", 0)] * 3 + fn[1:]
    pinned = {0: [("XP(0)", 8), ("RDD", 16)], 1: [("AXL(0)", 4), ("AXL(1)", 4)], 2: [("AXL(2)", 4), ("AXL(3)", 4)], a.xp1: [("XP(1)", 24)]}
    if head:
        pinned[3] = list(head)
    for ks in range(3):                                            # fragments of k-step ks+1: early in k-step ks
        si = next(i for i, s in enumerate(slots) if s[1] == ("k", ks, 2))
        pinned.setdefault(si, []).append((f"PF({ks + 1})", 16))

    # even spread: each queue is paid out at its own average rate (its cost / its gaps, --budget / --post-budget if given)
    npre = bar_slot + 1 - a.free_head
    npost = len(slots) - bar_slot - 1
    rate_pre = a.budget or sum(c for _, c in pre) / npre
    rate_post = a.post_budget or min(12.0, max(8.0, sum(c for _, c in post) / max(1, npost - 8)))
    lines, rep = [], []
    over = 0
    credit = 0.0
    for si, (mf, _) in enumerate(slots):
        items, used = [], 0
        for name, c in pinned.get(si, []):
            items.append(name); used += c
        queue, rate = (pre, rate_pre) if si <= bar_slot else (post, rate_post)
        if si == bar_slot + 1:
            credit = 0.0
        if si >= a.free_head:
            credit += rate
            while queue and (queue[0][1] <= credit + 2 or (si == bar_slot and queue is pre)):
                if queue[0][0].startswith("CI") and si <= a.xp1:   # the table offsets of the next step are not there yet
                    break
                name, c = queue.pop(0)
                if name == "-":                                    # filler: what follows waits for the next gap
                    credit = min(credit, 0.0)
                    break
                items.append(name); used += c; credit -= c
        if si == 0:
            items.insert(0, "ST(2)")
        if si == bar_slot:
            items += ["ST(0)", "BAR", "ST(1)"]
        over += max(0, used - 8)
        lines.append(f"{mf} " + " ".join(items) + (" " if items else "") + "GAP")
        rep.append((mf, used, items))
    left = [n for n, _ in pre + post if n != "-"]
    if left:                                                       # whatever is left goes behind the last MFMA (exposed)
        lines.append(" ".join(left) + " GAP")
    hdr = ["// generated by tools/gen_split2_schedule.py " + " ".join(f"--{k.replace('_', '-')} {v}" for k, v in sorted(vars(a).items())
                                                                     if k not in ("report", "o")),
           f"// {len(slots)} MFMA gaps; modelled issue-port cycles beyond the 8 free ones of each gap: {over}"]
    with open(a.o, "w") as fh:
        fh.write("\n".join(hdr + lines) + "\n")
    if a.report:
        for mf, used, items in rep:
            print(f"{mf:12s} {used:3d}  {' '.join(items)}")
    print(f"wrote {os.path.relpath(a.o)}: port cycles beyond the gaps {over}, behind the last MFMA: {len(left)} ops")


if __name__ == "__main__":
    main()
