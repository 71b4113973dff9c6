#!/usr/bin/env python3
"""Generate deepgrp_amd/csrc/gru_wave_phase_nu{1..4}.inc: the instruction order of one phase of gru_wave_kernel<NU> (gru_wave.hip),
the split-operand kernel of GRU models up to 64 units (16-unit granularity: NU = ceil(units / 16) unit groups, KS = ceil(NU / 2)
k-steps of 32).

A phase = the MFMAs (v_mfma_f32_16x16x32_f16) of tile X's step -- per k-step 9 NU recurrent ones (3 passes x 3 gates x NU unit groups)
and 3 of the Dense layer -- with tile Y's whole epilogue cut into single operations that sit in the gaps between them.  Both tiles
belong to the SAME wave and to no other: there is no barrier anywhere.  The macros are defined in gru_wave.hip:

    M_K(ks, n) M_D(ks, i)      X's MFMAs (asm volatile: stay in program order); PF(ks): X's fragments of k-step ks (LDS)
    XP(op)                     Y: base of its next step (0: read the byte, 1: table row offset)
    AXL(ug)                    Y: the candidate's input projection (table row) of unit group ug
    FS  FN(q, op)              Y: softmax + max-merge of the logits its last MFMA phase left in registers: FS folds the two strands'
                               partial tiles (two half-wave swaps + adds), then the chain 1..12 for q = 0, 1 (four windows each)
    G(e, op)                   Y: link `op` (0..11) of element e's gate chain (split_gate_op)
    CI(g, ug)                  Y: accumulator ug of gate g restarts from its table row
    PB(ug, op)                 Y: publish unit group ug: 0 state -> h, 1 hi = fp16(h), 2 h - hi, 3 lo = fp16(h - hi), 4 two LDS stores
    AV(ug, op)                 Y, attention pre-pass only: avg[t] of unit group ug (0: both strands' halves summed, 1: the store)
    RD0                        Y's next step: first fragments (behind the last publish)
    GAP                        sched_barrier(0)

Cost model as in tools/gen_split2_schedule.py (issue-port cycles: transcendental 8, plain VALU 4, LDS 4-8; a 16x16x32 MFMA holds the
port for 8 of its 16 cycles).  The epilogue does not fit the free cycles, so the stream is paced by the port and the aim is an even
spread.  Nothing that reads the previous phase's MFMA results goes into the first FREE_HEAD gaps.

    python tools/gen_wave_schedule.py [--nu 1 2 3 4] [--skew 3] [--report]
"""
import argparse
import os

T, P = 8, 4
GATE_T = {0, 1, 3, 5, 10}


def gate_pipeline(nu, skew):
    """Gate chains of the 4 NU elements, element e lagging e * skew links behind element 0; a unit group is published (and, in the
    attention pre-pass, its avg[t] stored) as soon as its last chain is done; its accumulators restart from the table rows as soon as
    its four chains have read them (link 4 is the last reader)."""
    ops, done = [], set()
    tau = 0
    ne = 4 * nu
    while len(done) < ne:
        for e in range(ne):
            op = tau - skew * e
            if 0 <= op < 12:
                ops.append((f"G({e}, {op})", T if op in GATE_T else P))
                if op == 4 and e % 4 == 3:
                    # "|": the restart waits for the next gap -- in the gap of the chains' last read the scheduler may hoist the LDS
                    # read above that read, the two generations of the accumulator then overlap and the allocator rotates registers
                    # with v_mov copies in front of MFMAs (tools/lint_split2_isa.py hazard A)
                    ops += [("|", 0)] + [(f"CI({g}, {e // 4})", 4) for g in range(3)]
                if op == 11:
                    done.add(e)
                    if e % 4 == 3:
                        g = e // 4
                        ops += [(f"PB({g}, 0)", 16), (f"AV({g}, 0)", 0), (f"PB({g}, 1)", 8), (f"PB({g}, 2)", 16), (f"AV({g}, 1)", 0),
                                (f"PB({g}, 3)", 8), (f"PB({g}, 4)", 8)]
        tau += 1
    return ops


def build(nu, skew, free_head, report):
    ks_n = (nu + 1) // 2
    slots = []
    for ks in range(ks_n):
        for n in range(9 * nu):
            slots.append(f"M_K({ks}, {n})")
        for i in range(3):
            slots.append(f"M_D({ks}, {i})")
    fn = [("FS", 16)]
    for q in range(2):
        fn += [(f"FN({q}, 1)", 8)] + [(f"FN({q}, {i})", 8) for i in range(2, 6)] + [(f"FN({q}, 6)", 12)] + \
              [(f"FN({q}, {i})", 8) for i in range(7, 11)] + [(f"FN({q}, 11)", 12), (f"FN({q}, 12)", 16)]
    queue = gate_pipeline(nu, skew)
    for i, item in enumerate(fn):                                  # softmax / merge links at even distances through the queue
        queue.insert(int((i + 0.5) * len(queue) / len(fn)) + i, item)
    queue.append(("RD0", 8))
    xp1 = min(free_head + 2, len(slots) - 2)
    pinned = {0: [("XP(0)", 4)], xp1: [("XP(1)", 16)]}
    for ug in range(nu):                                           # table rows of the step being finished: the head gaps
        pinned.setdefault(1 + ug // 2, []).append((f"AXL({ug})", 4))
    if ks_n > 1:
        pinned.setdefault(2, []).append(("PF(1)", 8))
    ngaps = len(slots) - free_head
    total = sum(c for _, c in queue)
    assert not any(n == "|" for n, _ in fn)
    rate = total / max(ngaps, 1)
    lines, rep = [], []
    credit, over = 0.0, 0
    for si, mf in enumerate(slots):
        items, used = [], 0
        for name, c in pinned.get(si, []):
            items.append(name); used += c
        if si >= free_head:
            credit += rate
            while queue and queue[0][1] <= credit + 2:
                if queue[0][0].startswith("CI") and si <= xp1:     # the table offset of the next step is not there yet
                    break
                name, c = queue.pop(0)
                if name == "|":
                    if items:
                        break                                      # what follows starts the next gap
                    continue
                items.append(name); used += c; credit -= c
        over += max(0, used - 8)
        lines.append(f"{mf} " + " ".join(items) + (" " if items else "") + "GAP")
        rep.append((mf, used, items))
    left = [n for n, _ in queue if n != "|"]
    if left:                                                       # whatever is left goes behind the last MFMA (exposed)
        lines.append(" ".join(left) + " GAP")
    hdr = [f"// generated by tools/gen_wave_schedule.py --nu {nu} --skew {skew} --free-head {free_head}",
           f"// {len(slots)} MFMA gaps; epilogue {total} modelled port cycles = {rate:.1f} per gap; beyond the 8 free ones of each gap: {over}"]
    if report:
        for mf, used, items in rep:
            print(f"{mf:12s} {used:3d}  {' '.join(items)}")
    return hdr + lines, over, len(left)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nu", type=int, nargs="*", default=[1, 2, 3, 4])
    ap.add_argument("--skew", type=int, default=3)
    ap.add_argument("--free-head", type=int, default=3)
    ap.add_argument("--report", action="store_true")
    ap.add_argument("--outdir", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "deepgrp_amd", "csrc"))
    a = ap.parse_args()
    for nu in a.nu:
        lines, over, left = build(nu, a.skew, a.free_head, a.report)
        path = os.path.join(a.outdir, f"gru_wave_phase_nu{nu}.inc")
        with open(path, "w") as fh:
            fh.write("\n".join(lines) + "\n")
        print(f"wrote {os.path.relpath(path)}: port cycles beyond the gaps {over}, behind the last MFMA: {left} ops")


if __name__ == "__main__":
    main()
