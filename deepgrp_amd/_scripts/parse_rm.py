#!/usr/bin/env python3
"""RepeatMasker output -> the six-column table deepgrp.preprocessing.preprocess_y reads
(contig, 0-based start, end, repeat number, repeat name, family).  Mirror of the reference's `parse_rm` console
script (deepgrp/_scripts/parse_rm.py); pinned by the reference's own fixture pair
tests/test_parse_rm_input.out -> tests/test_parse_rm_expect.bed (kept under tests/golden/).

Accepted line formats (anything else is skipped):
  * RepeatMasker `.out`: whitespace aligned, `score div del ins contig begin end (left) +|C name class/family ...`,
    1-based begin -> start = begin - 1;
  * UCSC `rmsk` table: tab separated, `bin score ... genoName genoStart genoEnd genoLeft +|- repName repClass
    repFamily`, already 0-based; family = class when both agree, else `class/family`.
Only ten repeat families are numbered (1..10, table below); a row is written when its family -- or failing that
its repeat name -- is one of them.  Simple repeats / satellites whose unit is built from the HSATII pentamer GGAAT
(any rotation or strand, at most single-base variants next to at least one exact copy) count as HSATII."""
import argparse
import re
import sys
from typing import Dict, Iterable, Iterator, List, NamedTuple, Optional, Tuple

REPEATS: Tuple[str, ...] = ("HSATII", "ALR/Alpha", "SINE/Alu", "LINE/L1", "SINE/MIR", "LINE/L2", "LTR/ERV1", "LTR/ERVL",
                            "LTR/ERVL-MaLR", "LTR/Gypsy")
NUMBER: Dict[str, int] = {name: i for i, name in enumerate(REPEATS, 1)}
PENTAMER = "GGAAT"

_OUT_LINE = re.compile(r"^\s*\d+\s+\S+\s+\S+\s+\S+\s+(\S+)\s+(\d+)\s+(\d+)\s+\S+\s+[+C]\s+(\S+)\s+(\S+)")
_RMSK_LINE = re.compile(r"^\d+(\t\d+){4}\t(\S+)\t(\d+)\t(\d+)\t\S+\t[+-]\t(\S+)\t(\S+)\t(\S+)")
_UNIT = re.compile(r"^\(([ACGT]+)\)n")
_COMPLEMENT = str.maketrans("ACGT", "TGCA")


class Repeat(NamedTuple):
    ctg: Optional[str]
    start: Optional[int]
    end: Optional[int]
    typ: int
    rep: str
    fam: Optional[str]

    def __str__(self) -> str:
        return f"{self.ctg}\t{self.start}\t{self.end}\t{self.typ}\t{self.rep}\t{self.fam}"


def pentamer_sets(unit: str = PENTAMER) -> Tuple[Dict[str, int], Dict[str, int]]:
    """(exact, one_off): the unit, its reverse complement and every rotation of the two; and all words one
    substitution away from one of those."""
    words: List[str] = [unit, unit[::-1].translate(_COMPLEMENT)]
    words += [w[j:] + w[:j] for w in list(words) for j in range(1, len(w))]
    exact = {w: k for k, w in enumerate(words)}
    one_off: Dict[str, int] = {}
    for w in words:
        for i, c in enumerate(w):
            for b in "ACGT":
                if b != c:
                    one_off[w[:i] + b + w[i + 1:]] = 1
    return exact, one_off


def parse_line(line: str) -> Repeat:
    """Fields of one line (all None / 0 when the line has neither format)."""
    ctg = start = end = fam = None
    rep = ""
    m = _OUT_LINE.match(line)
    if m:
        ctg, start, end, rep, fam = m.group(1), int(m.group(2)) - 1, int(m.group(3)), m.group(4), m.group(5)
    else:
        m = _RMSK_LINE.match(line)
        if m:
            ctg, start, end, rep = m.group(2), int(m.group(3)), int(m.group(4)), m.group(5)
            fam = m.group(6) if m.group(6) == m.group(7) else m.group(6) + "/" + m.group(7)
    typ = NUMBER.get(fam, 0) or NUMBER.get(rep, 0)
    return Repeat(ctg, start, end, typ, rep, fam)


def _is_hsat2_unit(unit: str, exact: Dict[str, int], one_off: Dict[str, int]) -> bool:
    if unit in exact:
        return True
    k = len(PENTAMER)
    if len(unit) % k:
        return False
    parts = [unit[j:j + k] for j in range(0, len(unit), k)]
    hits = sum(p in exact for p in parts)
    near = sum(p not in exact and p in one_off for p in parts)
    return hits > 0 and (hits + near) * k == len(unit)


def read_repeatmasker(exact: Dict[str, int], one_off: Dict[str, int], lines: Iterable[str]) -> Iterator[Repeat]:
    """The numbered repeats of a RepeatMasker listing, in input order."""
    for line in lines:
        r = parse_line(line)
        if r.typ == 0 and r.fam in ("Simple_repeat", "Satellite"):
            m = _UNIT.match(r.rep)
            if m and _is_hsat2_unit(m.group(1), exact, one_off):
                r = r._replace(typ=NUMBER["HSATII"])
        if r.ctg and r.typ > 0:
            yield r


def main(argv: Optional[List[str]] = None) -> None:
    ap = argparse.ArgumentParser(description="Reads Repeatmasker output to bed file (not all repeats!!)")
    ap.add_argument("file", type=argparse.FileType("r"), help="Repeatmasker output")
    ap.add_argument("-o", "--outputfile", type=str, default=None, help="Output filename")
    args = ap.parse_args(argv)
    exact, one_off = pentamer_sets()
    rows = (str(r) for r in read_repeatmasker(exact, one_off, args.file))
    if args.outputfile:
        with open(args.outputfile, "w") as fh:
            for row in rows:
                fh.write(row + "\n")
    else:
        for row in rows:
            sys.stdout.write(row + "\n")


if __name__ == "__main__":
    main()
