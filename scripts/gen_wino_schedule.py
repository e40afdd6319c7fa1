#!/usr/bin/env python
"""Generates megapose6d_amd/csrc/conv_wino_bf16_sched.h: the slot tables of the bf16x9 Winograd K loop (csrc/conv_wino_bf16.hip).

A frequency point of the loop is 36 MFMAs of one wave per SIMD; everything else the wave has to do -- split the NEXT point's V fragments into
bf16 pieces, transform the next step's input patch, request weights / patch rows, read fragments -- rides in the 36 gaps behind them.  Round 6
measured what a gap can hide (scripts/microbench/mfma_gap_fillers.hip, profiles/r06_wino_kloop_experiments.txt):
  * up to 5 independent VALU instructions: ~free (33.7 -> 35.5 cycles per MFMA); the 6th costs 2, the 7th 5, the 8th 4.5 cycles;
  * a ds_write_b128 costs 20 cycles whatever else is in the gap, and up to 6 VALU instructions ride free in its shadow.
The round-4/5 tables packed the whole split into the first 16 gaps (6 - 8 VALU instructions in most of them) and left the last 20 nearly
empty.  These tables spread it: one 4-instruction group per gap (split groups in the even gaps, transform groups in the odd ones), the v_perm
pairs in the ds_write gaps (4 + 2 VALU + the write = what the write's shadow hides) or as single instructions beside a group (4 + 1).

The script CHECKS what the tables must respect and fails otherwise:
  budget      per gap: <= 5 instructions, or <= 7 VALU + the ds_write in a gap that holds one;
  dependences of the split (one temporary set sm / sr / sq shared by the four (tile block, half) chains):
              A(k) < B(k) < C(k) < D(k) < A(k+1);  pa(k) before raw(k) is re-read and after it was read;  pb(k) in (B(k), B(k+1));
              pc(k) in (D(k), D(k+1));  a raw re-read R(k) after A(k), B(k), pa(k);
  of the transform: T(row, 0..3) < P(0) ; P(c) < W(c) < P(c+1) ; W(3) of row 0 < T(row 1, .) ; P(3) of row 0 < T(row 1, .).
Run: python scripts/gen_wino_schedule.py   (rewrites the header; --check: exit 1 if the header on disk is stale)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "megapose6d_amd" / "csrc" / "conv_wino_bf16_sched.h"
CHAINS = [(0, 0), (0, 1), (1, 0), (1, 1)]   # (tile block I, half H) of the raw fragment a chain splits
U_ORDER = [2, 5, 1, 4, 0, 3]                # weight fragments [cout block][piece]: the pieces the next point multiplies first are requested first


def split_groups():
    out = []
    for k in range(4):
        out += [("A", k), ("B", k), ("C", k), ("D", k)]
    return out


def block(kind):
    """-> list of 36 lists of atoms.  kind: 'TR' (transform rows R0, R1), 'PL' (patch rows R0, R1 requested), 'RD' (last step, points 0 / 1:
    nothing but the split, the weights and the fragment reads), 'RES' (last step, point 2: residual requests 0..7 instead of patch requests,
    no fragment reads)."""
    slots = [[] for _ in range(36)]
    sg = split_groups()
    for i, g in enumerate(sg):
        slots[2 * i].append(g)
    if kind == "TR":
        tg = [("T", 0, b) for b in range(4)] + [("P", 0, c) for c in range(4)] + [("T", 1, b) for b in range(4)] + [("P", 1, c) for c in range(4)]
        for i, g in enumerate(tg):
            slots[2 * i + 1].append(g)
        for row, first in ((0, 10), (1, 26)):   # W(row, c) one gap behind P(row, c)
            for c in range(4):
                slots[first + 2 * c].append(("W", row, c))
        # v_perm pairs in the ds_write gaps, in front of the group of that gap where they read what it overwrites
        for s, atom, front in ((10, ("pb", 0), True), (12, ("pb", 1), False), (14, ("pc", 0), True), (16, ("pc", 1), False),
                               (26, ("pb", 2), True), (28, ("pc", 2), False), (30, ("pb", 3), False), (32, ("pc", 3), False)):
            pair = [(atom[0], atom[1], 0), (atom[0], atom[1], 1)]
            slots[s][:] = (pair + slots[s]) if front else (slots[s] + pair)
        for j, k in enumerate(U_ORDER):
            slots[1 + 2 * j].append(("U", k))
        for k in range(4):   # pa(k) as two single instructions beside transform groups
            slots[13 + 4 * k].append(("pa", k, 0))
            slots[15 + 4 * k].append(("pa", k, 1))
        for k, s in zip(range(4), (29, 31, 33, 34)):
            slots[s].append(("R", k))
    else:
        for k, s in zip(range(4), (1, 9, 17, 25)):
            slots[s] += [("pa", k, 0), ("pa", k, 1)]
        for k, s in zip(range(4), (3, 11, 19, 27)):
            slots[s] += [("pb", k, 0), ("pb", k, 1)]
        for k, s in zip(range(4), (7, 15, 23, 31)):
            slots[s] += [("pc", k, 0), ("pc", k, 1)]
        for j, k in enumerate(U_ORDER):
            slots[1 + 2 * j].append(("U", k))
        if kind == "PL":
            for j in range(8):
                slots[13 + 2 * j].append(("L", j // 4, j % 4))
        if kind == "RES":
            for j in range(8):
                slots[13 + 2 * j].append(("X", j))
        if kind in ("PL", "RD"):
            for k, s in zip(range(4), (5, 13, 21, 29)):
                slots[s].append(("R", k))
    return slots


def n_instr(atom):
    return 4 if atom[0] in "ABCDTP" else 1


def check(kind, slots):
    pos = {}
    for s, atoms in enumerate(slots):
        for i, a in enumerate(atoms):
            assert a not in pos, (kind, "twice", a)
            pos[a] = (s, i)
        n = sum(n_instr(a) for a in atoms)
        has_w = any(a[0] == "W" for a in atoms)
        assert n <= (8 if has_w else 5), (kind, "gap", s, "holds", n, atoms)
    before = lambda a, b: pos[a] < pos[b]   # noqa: E731
    for k in range(4):
        assert before(("A", k), ("B", k)) and before(("B", k), ("C", k)) and before(("C", k), ("D", k)), (kind, "chain", k)
        assert pos[("B", k)][0] > pos[("A", k)][0] and pos[("C", k)][0] > pos[("B", k)][0] and pos[("D", k)][0] > pos[("C", k)][0], \
            (kind, "dependent groups of chain", k, "share a gap")
        if k < 3:
            assert before(("D", k), ("A", k + 1)), (kind, "chain order", k)
        for j in (0, 1):
            assert before(("B", k), ("pb", k, j)) and pos[("pb", k, j)][0] >= pos[("B", k)][0] + 1, (kind, "pb after B", k)
            assert before(("D", k), ("pc", k, j)) and pos[("pc", k, j)][0] >= pos[("D", k)][0] + 1, (kind, "pc after D", k)
            if k < 3:
                assert before(("pb", k, j), ("B", k + 1)), (kind, "pb before the next B", k)
                assert before(("pc", k, j), ("D", k + 1)), (kind, "pc before the next D", k)
            if ("R", k) in pos:
                assert before(("pa", k, j), ("R", k)), (kind, "pa before the re-read", k)
        if ("R", k) in pos:
            assert before(("B", k), ("R", k)) and before(("A", k), ("R", k)), (kind, "re-read after the chain's reads", k)
    if kind == "TR":
        for row in (0, 1):
            for b in range(4):
                assert before(("T", row, b), ("P", row, 0)), (kind, "T before P", row, b)
            for c in range(4):
                assert before(("P", row, c), ("W", row, c)) and pos[("W", row, c)][0] > pos[("P", row, c)][0], (kind, "W one gap behind P", row, c)
                if c < 3:
                    assert before(("W", row, c), ("P", row, c + 1)), (kind, "W before the next P", row, c)
        for b in range(4):
            assert before(("P", 0, 3), ("T", 1, b)) and before(("W", 0, 3), ("P", 1, 0)), (kind, "rows in order")
    n_u = sum(1 for a in pos if a[0] == "U")
    assert n_u == 6, (kind, "weight requests", n_u)
    return pos


def atom_text(a, kind):
    t = a[0]
    if t == "A":
        return f"WB_SPA({CHAINS[a[1]][0]}, {CHAINS[a[1]][1]})"
    if t == "B":
        return f"WB_SP1({CHAINS[a[1]][0]}, {CHAINS[a[1]][1]})"
    if t == "C":
        return "WB_SPC()"
    if t == "D":
        return "WB_SP3()"
    if t in ("pa", "pb", "pc"):
        return f"WB_PRM_{t[1].upper()}(AN, {CHAINS[a[1]][0]}, {CHAINS[a[1]][1]}, {a[2]})"
    if t == "U":
        return f"WB_LOAD_U1(UN, NST, NFI, {a[1]})"
    if t == "T":
        return f"WB_TR_T(R{a[1]}, {a[2]})"
    if t == "P":
        return f"WB_TR_P({a[2]})"
    if t == "W":
        return f"WB_TR_W(R{a[1]}, {a[2]}, VW)"
    if t == "L":
        return f"WB_LOAD_PATCH1(R{a[1]}, {a[2]}, CS)"
    if t == "R":
        return f"WB_READ_RAW1(VBN, N2FI, {a[1]})"
    if t == "X":
        return f"WB_LOAD_RES1({a[1]})"
    raise KeyError(a)


ARGS = {"TR": "AC, AN, UC, UN, NST, NFI, R0, R1, VW, VBN, N2FI", "PL": "AC, AN, UC, UN, NST, NFI, R0, R1, CS, VBN, N2FI",
        "RD": "AC, AN, UC, UN, NST, NFI, VBN, N2FI", "RES": "AC, AN, UC, UN, NST, NFI"}
WHAT = {"TR": "points 0 / 1 of a step: rows R0, R1 of the NEXT step's input transform", "PL": "points 2 / 3 of a step: rows R0, R1 of the patch of the step after next are requested",
        "RD": "points 0 / 1 of the LAST step: split, weights and fragment reads only", "RES": "point 2 of the LAST step: residual requests 0..7 in place of the patch requests"}


def render():
    out = ["// GENERATED by scripts/gen_wino_schedule.py -- do not edit by hand (the script checks the budgets and dependences it documents).",
           "// Slot tables of the bf16x9 Winograd K loop: 36 MFMAs per frequency point, the other work of the wave in the gaps behind them.",
           "// Included inside conv3x3_wino_bf16x9 (csrc/conv_wino_bf16.hip), which defines the WB_* primitives used here.", ""]
    for kind in ("TR", "PL", "RD", "RES"):
        slots = block(kind)
        check(kind, slots)
        n_fill = sum(n_instr(a) for s in slots for a in s)
        out.append(f"// {WHAT[kind]} ({n_fill} instructions in the gaps; at most "
                   f"{max(sum(n_instr(a) for a in s) for s in slots)} in one)")
        out.append(f"#define WB_BLK_{kind}({ARGS[kind]}) \\")
        lines, run = [], []
        for s, atoms in enumerate(slots):
            if atoms:
                if run:
                    lines.append("  " + " ".join(run) + " WB_SB")
                    run = []
                lines.append(f"  WB_M({s}, AC, UC) WB_SB " + " ".join(atom_text(a, kind) for a in atoms) + " WB_SB")
            else:
                run.append(f"WB_M({s}, AC, UC)")
        if run:
            lines.append("  " + " ".join(run) + " WB_SB")
        out.append(" \\\n".join(lines))
        out.append("")
    return "\n".join(out)


if __name__ == "__main__":
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if OUT.is_file() and OUT.read_text() == text else 1)
    OUT.write_text(text)
    print(f"wrote {OUT}")
