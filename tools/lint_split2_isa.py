#!/usr/bin/env python3
"""Audit the ISA of gru_split2_kernel for the two hazards its inline-asm MFMAs hide from hipcc.

gru_split2.hip issues every MFMA as an `asm volatile` statement (weights pinned in AGPRs).  hipcc does not know those statements
are MFMAs, so it pads neither of these pairs (cdna_hip_programming.md 5.7 item 2):

  A  a VALU write of a VGPR  ->  an MFMA reading it as A, B or C:            2 wait states
  B  an MFMA's D             ->  any other reader or writer of it:           12 wait states (8-pass XDL), except an MFMA of the
                                                                             same shape taking it whole as C, or overwriting it

The source keeps them apart by construction (operands come from LDS reads, results are read a phase later), but the register
allocator is free to put a copy (`v_mov`) of an accumulator or a fragment next to an MFMA -- on a loop edge, or where a live
range was split -- and then the kernel computes with stale registers on every launch, without a fault (seen with a schedule that
left nothing but four LDS reads between a phase's barrier and the loop's back edge: `v_mov_b64 v[88:89], v[36:37]` directly in
front of the MFMA accumulating into v[88:91]).  So the build runs this audit on the compiler's own output and fails on a hit:

    python tools/lint_split2_isa.py <file.s> [--kernel gru_split2_kernel] [--fix <patched.s>]

Both are audited along the fall-through order AND across every branch (the instructions in front of an s_branch / s_cbranch to the
instructions behind its target label: a taken branch skips what the text order counts as wait states).  With --fix the missing wait
states are written into a copy of the file as s_nop in front of the later instruction of each pair (the build of gru_wave.o assembles
that copy: the allocator's copies differ from one schedule to the next, a few s_nop at the places that need them cost nothing
measurable), and the copy is audited again; the exit status is that of the copy.

A wait state = one issued instruction (`s_nop N` = N + 1), counted the way LLVM's hazard recognizer does.

Two more things are audited because they once cost the kernel a fifth of its time without changing a result (item 4 of the same
section): more than 16 `v_accvgpr_read` / `v_accvgpr_write` in a kernel (the weights are born in AGPRs and every MFMA names them there;
the round-1 build spent 338 copies per wave-step; a handful is the allocator parking a value in one of the 56 free AGPRs) and any scratch (`.amdhsa_private_segment_fixed_size` / `.vgpr_spill_count` not 0).
"""
import argparse
import os
import re
import sys

REG = re.compile(r"(?<![\w.])([va])(?:\[(\d+):(\d+)\]|(\d+))(?![\w\[])")
STATES_VALU_TO_MFMA = 2
STATES_MFMA_TO_USE = 12
ACC_COPIES_ALLOWED = 16


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        lo = int(m.group(2) if m.group(2) is not None else m.group(4))
        hi = int(m.group(3) if m.group(3) is not None else m.group(4))
        out.update((m.group(1), i) for i in range(lo, hi + 1))
    return out


class Ins:
    __slots__ = ("line", "text", "mn", "ops", "states", "is_mfma", "wr", "rd", "label", "target")

    def __init__(self, line, text):
        self.line, self.text = line, text
        self.label = self.target = None
        if text.endswith(":"):                                     # a label: no instruction, no wait state
            self.label, self.mn, self.ops, self.states, self.is_mfma, self.wr, self.rd = text[:-1], "", [], 0, False, set(), set()
            return
        parts = text.split(None, 1)
        self.mn = parts[0]
        self.ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        self.is_mfma = self.mn.startswith("v_mfma") or self.mn.startswith("v_smfma")
        self.states = 1
        if self.mn == "s_nop":
            self.states = int(self.ops[0], 0) + 1
        if self.mn.startswith(("s_branch", "s_cbranch")) and self.ops:
            self.target = self.ops[0]
        self.wr, self.rd = set(), set()
        vec_alu = self.mn.startswith("v_")
        load = self.mn.startswith(("ds_read", "global_load", "buffer_load", "flat_load", "scratch_load"))
        if self.is_mfma:
            self.wr = regs(self.ops[0])
            for o in self.ops[1:4]:
                self.rd |= regs(o)
        elif vec_alu or load:
            if self.ops:
                self.wr = regs(self.ops[0])
            for o in self.ops[1:]:
                self.rd |= regs(o)
            if vec_alu and (self.mn.startswith(("v_fmac", "v_mac", "v_fma_mix", "v_dot")) or "dpp" in self.mn or "op_sel" in text):
                self.rd |= self.wr                                 # accumulating / partially written destinations
        else:                                                      # stores, atomics, exports: every register is read
            for o in self.ops:
                self.rd |= regs(o)


def kernels(path, pattern):
    name, body = None, []
    with open(path) as fh:
        for no, raw in enumerate(fh, 1):
            s = raw.split(";")[0].rstrip()
            if not s.strip():
                continue
            m = re.match(r"^(\S+):\s*$", s)
            if m and not s.startswith("\t") and not m.group(1).startswith("."):
                if name and body:
                    yield name, body
                name, body = (m.group(1) if pattern in m.group(1) else None), []
                continue
            if name is not None and re.match(r"^\.L\w+:\s*$", s):
                body.append(Ins(no, s.strip()))
                continue
            if name is None or not s.startswith("\t"):
                continue
            t = s.strip()
            if t.startswith(".") and not t.endswith(":"):
                continue
            body.append(Ins(no, t))
            if t.startswith("s_endpgm"):
                yield name, body
                name, body = None, []
    if name and body:
        yield name, body


def audit_seq(seq, seam=None):
    """Hazards along one linear instruction sequence.  With `seam` = index of the first instruction behind a taken branch only the
    pairs that straddle it are reported (the others belong to the fall-through walk)."""
    hits = []
    for i, ins in enumerate(seq):
        if not ins.is_mfma:
            continue
        # A: VALU write -> this MFMA's operands
        states, j = 0, i - 1
        while j >= 0 and states < STATES_VALU_TO_MFMA:
            p = seq[j]
            if p.mn.startswith("v_") and not p.is_mfma and p.wr & ins.rd and (seam is None or j < seam <= i):
                hits.append(("A", p, ins, states))
            states += p.states
            j -= 1
        # B: this MFMA's D -> readers / writers
        states, j = 0, i + 1
        while j < len(seq) and states < STATES_MFMA_TO_USE:
            q = seq[j]
            touched = (q.rd | q.wr) & ins.wr
            if touched:
                # an MFMA that takes the result WHOLE as its C operand may follow back to back, whatever its own destination (LLVM
                # GCNHazardRecognizer::checkMAIHazards90A, "FullReg": zero wait states for same-latency XDL ops), and so may a same-shape
                # MFMA that merely overwrites it (in-order pipe, equal latency).  Reading it as A or B, or a part of it as C, is a hazard.
                ab = (regs(q.ops[1]) | regs(q.ops[2])) & ins.wr if q.is_mfma else set()
                c = regs(q.ops[3]) & ins.wr if q.is_mfma and len(q.ops) > 3 else set()
                chain = q.is_mfma and q.mn == ins.mn and not ab and (not c or regs(q.ops[3]) == ins.wr)
                if not chain and (seam is None or i < seam <= j):
                    hits.append(("B", ins, q, states))
                if q.wr >= ins.wr:                                 # overwritten whole: later uses belong to the new value
                    break
            states += q.states
            j += 1
    return hits


def audit(body):
    """Fall-through order, then every branch: the instructions in front of it joined to those behind its target label."""
    hits = audit_seq(body)
    where = {ins.label: k for k, ins in enumerate(body) if ins.label}
    span = STATES_MFMA_TO_USE + 4
    for b, ins in enumerate(body):
        if ins.target is None or ins.target not in where:
            continue
        t = where[ins.target]
        head = body[max(0, b - span):b + 1]
        hits += audit_seq(head + body[t:t + span], seam=len(head))
    return hits


def fix(path, out, pattern, rounds=6):
    """Copy `path` to `out` with the missing wait states of the matching kernels inserted as s_nop in front of the second instruction
    of each pair; repeated until the copy is clean (an insertion moves later pairs apart, never together)."""
    import shutil
    shutil.copyfile(path, out)
    for _ in range(rounds):
        need = {}
        for _name, body in kernels(out, pattern):
            for kind, first, second, states in audit(body):
                want = (STATES_VALU_TO_MFMA if kind == "A" else STATES_MFMA_TO_USE) - states
                need[second.line] = max(need.get(second.line, 0), want)
        if not need:
            return True
        lines = open(out).read().split("\n")
        for no in sorted(need, reverse=True):
            k = need[no]
            pad = []
            while k > 0:
                pad.append(f"\ts_nop {min(k, 16) - 1}                                 ; wait states for an inline-asm MFMA (lint_split2_isa.py --fix)")
                k -= 16
            lines[no - 1:no - 1] = pad
        with open(out, "w") as fh:
            fh.write("\n".join(lines))
    return False


def spills(path, pattern):
    """(kernel, key, value) for every non-zero scratch / spill entry of the matching kernels' descriptors and metadata."""
    out, cur = [], None
    with open(path) as fh:
        for raw in fh:
            t = raw.strip()
            m = re.match(r"\.amdhsa_kernel\s+(\S+)", t) or re.match(r"\.name:\s+(\S+)", t)
            if m:
                cur = m.group(1) if pattern in m.group(1) else None
                continue
            m = re.match(r"\.?(amdhsa_private_segment_fixed_size|private_segment_fixed_size:|vgpr_spill_count:|sgpr_spill_count:)\s+(\d+)", t)
            if m and cur and int(m.group(2)) != 0:
                out.append((cur, m.group(1).rstrip(":"), int(m.group(2))))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("--kernel", default="gru_split2_kernel")
    ap.add_argument("--fix", metavar="OUT", help="write a copy with the missing wait states inserted, and audit that copy")
    ap.add_argument("--max-acc-copies", type=int, default=ACC_COPIES_ALLOWED,
                    help="v_accvgpr_read / _write tolerated per kernel (a kernel that fills its 256 architectural VGPRs lets the allocator park "
                         "values in free AGPRs: a few dozen per step are noise; hundreds are the weights being copied, the round-1 pathology)")
    a = ap.parse_args()
    if a.fix:
        before = sum(len(audit(body)) for _n, body in kernels(a.asm, a.kernel))
        if not fix(a.asm, a.fix, a.kernel):
            print(f"{a.fix}: still not clean after patching")
        print(f"{a.asm}: {before} hazard(s) patched into {a.fix}")
        a.asm = a.fix
    total = nk = 0
    for kern, key, val in spills(a.asm, a.kernel):
        total += 0 if os.environ.get("DGRP_LINT_ALLOW_SCRATCH") else 1          # (timing experiments only: the shipped build never sets it)
        print(f"{kern}: {key} = {val} (the kernel must not touch scratch)")
    for name, body in kernels(a.asm, a.kernel):
        nk += 1
        hits = audit(body)
        total += len(hits)
        nm = sum(1 for x in body if x.is_mfma)
        body_n = sum(1 for x in body if not x.label)
        acc = [x for x in body if x.mn.startswith("v_accvgpr")]
        if acc:                                                    # a handful = the allocator parking a value in a free AGPR; the pathology is hundreds
            bad = len(acc) > a.max_acc_copies
            total += 1 if bad else 0
            print(f"{name}: {len(acc)} v_accvgpr copies{' (more than ' + str(a.max_acc_copies) + ')' if bad else ' (tolerated)'}, first: line {acc[0].line}: {acc[0].text}")
        print(f"{name}: {body_n} instructions, {nm} MFMAs, {len(hits)} hazard(s)")
        for kind, first, second, states in hits[:12]:
            need = STATES_VALU_TO_MFMA if kind == "A" else STATES_MFMA_TO_USE
            print(f"  {kind}: line {first.line}: {first.text}\n     line {second.line}: {second.text}\n     {states} wait state(s) between, {need} needed")
    if nk == 0:
        print(f"no kernel matching {a.kernel!r} in {a.asm}")
        return 2
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
