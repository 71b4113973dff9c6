#!/usr/bin/env python3
"""Audit the ISA of gru_split2_kernel for the two hazards its inline-asm MFMAs hide from hipcc.

gru_split2.hip issues every MFMA as an `asm volatile` statement (weights pinned in AGPRs).  hipcc does not know those statements
are MFMAs, so it pads neither of these pairs (cdna_hip_programming.md 5.7 item 2):

  A  a VALU write of a VGPR  ->  an MFMA reading it as A, B or C:            2 wait states
  B  an MFMA's D             ->  any other reader or writer of it:           12 wait states (8-pass XDL), except the next
                                                                             MFMA of an accumulate chain taking it whole as C

The source keeps them apart by construction (operands come from LDS reads, results are read a phase later), but the register
allocator is free to put a copy (`v_mov`) of an accumulator or a fragment next to an MFMA -- on a loop edge, or where a live
range was split -- and then the kernel computes with stale registers on every launch, without a fault (seen with a schedule that
left nothing but four LDS reads between a phase's barrier and the loop's back edge: `v_mov_b64 v[88:89], v[36:37]` directly in
front of the MFMA accumulating into v[88:91]).  So the build runs this audit on the compiler's own output and fails on a hit:

    python tools/lint_split2_isa.py <file.s> [--kernel gru_split2_kernel]

A wait state = one issued instruction (`s_nop N` = N + 1), counted the way LLVM's hazard recognizer does.

Two more things are audited because they once cost the kernel a fifth of its time without changing a result (item 4 of the same
section): more than 16 `v_accvgpr_read` / `v_accvgpr_write` in a kernel (the weights are born in AGPRs and every MFMA names them there;
the round-1 build spent 338 copies per wave-step; a handful is the allocator parking a value in one of the 56 free AGPRs) and any scratch (`.amdhsa_private_segment_fixed_size` / `.vgpr_spill_count` not 0).
"""
import argparse
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
    __slots__ = ("line", "text", "mn", "ops", "states", "is_mfma", "wr", "rd")

    def __init__(self, line, text):
        self.line, self.text = line, text
        parts = text.split(None, 1)
        self.mn = parts[0]
        self.ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        self.is_mfma = self.mn.startswith("v_mfma") or self.mn.startswith("v_smfma")
        self.states = 1
        if self.mn == "s_nop":
            self.states = int(self.ops[0], 0) + 1
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
            if name is None or not s.startswith("\t"):
                continue
            t = s.strip()
            if t.startswith(".") or t.endswith(":"):
                continue
            body.append(Ins(no, t))
            if t.startswith("s_endpgm"):
                yield name, body
                name, body = None, []
    if name and body:
        yield name, body


def audit(body):
    hits = []
    for i, ins in enumerate(body):
        if not ins.is_mfma:
            continue
        # A: VALU write -> this MFMA's operands
        states, j = 0, i - 1
        while j >= 0 and states < STATES_VALU_TO_MFMA:
            p = body[j]
            if p.mn.startswith("v_") and not p.is_mfma and p.wr & ins.rd:
                hits.append(("A", p, ins, states))
            states += p.states
            j -= 1
        # B: this MFMA's D -> readers / writers
        states, j = 0, i + 1
        while j < len(body) and states < STATES_MFMA_TO_USE:
            q = body[j]
            touched = (q.rd | q.wr) & ins.wr
            if touched:
                chain = q.is_mfma and regs(q.ops[3]) == ins.wr and q.wr == ins.wr and not (regs(q.ops[1]) | regs(q.ops[2])) & ins.wr
                if not chain:
                    hits.append(("B", ins, q, states))
                if q.wr >= ins.wr:                                 # overwritten whole: later uses belong to the new value
                    break
            states += q.states
            j += 1
    return hits


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
    a = ap.parse_args()
    total = nk = 0
    for kern, key, val in spills(a.asm, a.kernel):
        total += 1
        print(f"{kern}: {key} = {val} (the kernel must not touch scratch)")
    for name, body in kernels(a.asm, a.kernel):
        nk += 1
        hits = audit(body)
        total += len(hits)
        nm = sum(1 for x in body if x.is_mfma)
        acc = [x for x in body if x.mn.startswith("v_accvgpr")]
        if acc:                                                    # a handful = the allocator parking a value in a free AGPR; the pathology is hundreds
            bad = len(acc) > ACC_COPIES_ALLOWED
            total += 1 if bad else 0
            print(f"{name}: {len(acc)} v_accvgpr copies{' (more than ' + str(ACC_COPIES_ALLOWED) + ')' if bad else ' (tolerated)'}, first: line {acc[0].line}: {acc[0].text}")
        print(f"{name}: {len(body)} instructions, {nm} MFMAs, {len(hits)} hazard(s)")
        for kind, first, second, states in hits[:12]:
            need = STATES_VALU_TO_MFMA if kind == "A" else STATES_MFMA_TO_USE
            print(f"  {kind}: line {first.line}: {first.text}\n     line {second.line}: {second.text}\n     {states} wait state(s) between, {need} needed")
    if nk == 0:
        print(f"no kernel matching {a.kernel!r} in {a.asm}")
        return 2
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
