#!/usr/bin/env python
"""Static hazard lint of the gfx950 machine code of a kernel that mixes inline asm with compiler-scheduled code (no GPU needed).

    python scripts/isa_hazards.py megapose6d_amd/csrc/conv_wino_bf16.hip conv3x3_wino_bf16x9ILi0E [-DMP_WINO_PERMUTE=3 ...]

Why: hipcc treats an `asm volatile` statement as ONE opaque instruction -- it allocates the operands, but its hazard recogniser neither pads
a dependency whose producer or consumer sits inside the string nor knows what the string writes when.  conv3x3_wino_bf16x9 (the kernel that
is 60 % of every pose-pipeline step) places its split / transform arithmetic as inline asm under builtin MFMAs and resets its accumulators
with inline-asm MFMAs.  Round 5 saw that kernel return inf / NaN whenever an edit elsewhere made the register allocator choose differently;
round 6 found the cause with the first version of this script: the compiler materialised the zero operand of the reset MFMAs with `v_mov`
DIRECTLY in front of the asm statement (0 wait states; the hardware needs 2 between a VALU write and an MFMA reading the register as A / B),
so the first MFMA multiplied the registers' PREVIOUS contents.  This lint turns "validated for one register assignment" into rules that
hold for any assignment; tests/test_wino_isa_hazards_cpu.py runs it on the product build and on deliberately permuted builds.

Rules (wait states: every instruction issued in between counts 1, `s_nop N` counts N + 1; the loop bodies are walked twice so that the
back edge is covered; the walk is linear -- conditional forward branches are treated as not taken, which only adds checks; of an if / else the `then`
side is walked):
  R1  VALU write of a VGPR -> MFMA reading it as SrcA / SrcB / SrcC: >= 2 wait states.
  R2  a register that is the destination of a load still in flight (vmcnt / lgkmcnt modelled in order, exactly as `s_waitcnt` counts them)
      is neither read nor written before the wait that covers the load.
  R3  inside one asm statement no instruction reads a register an EARLIER instruction of the same statement wrote unless the source text
      asks for it (our statements never do: a hit means a missing early-clobber `&`).
  R4  the destination of an MFMA is not touched by a non-MFMA instruction within 18 wait states (8-pass MFMA), and not by a later MFMA as
      a partially overlapping SrcC.
  R5  a store's data registers (> 64 bit, buffer / global / scratch) are not overwritten by a VALU instruction right behind it (1 wait state).
"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{ROOT / 'include'}", "-Wno-unused-function"]
MFMA_GUARD_STATES = 18


def _regs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return [m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    return [tok] if re.fullmatch(r"[va]\d+", tok) else []


def compile_kernel(src, kernel_substr, extra=()):
    asm = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", "-o", "-", str(src)], capture_output=True,
                         text=True, check=True).stdout
    parts = re.split(r"\n(_Z\w+):[^\n]*\n", asm)
    for i in range(1, len(parts), 2):
        if kernel_substr in parts[i]:
            return parts[i + 1].split("s_endpgm")[0]
    raise KeyError(kernel_substr)


def compile_all(src, extra=()):
    """-> {mangled kernel name: its assembly text} of every kernel in the translation unit (one compilation)"""
    asm = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", "-o", "-", str(src)], capture_output=True,
                         text=True, check=True).stdout
    parts = re.split(r"\n(_Z\w+):[^\n]*\n", asm)
    return {parts[i]: parts[i + 1].split("s_endpgm")[0] for i in range(1, len(parts), 2)}


def parse(text):
    """-> list of {op, ops, asm (statement id or None), text, label}"""
    out, stmt, n_stmt = [], None, 0
    for raw in text.splitlines():
        s = raw.strip()
        if s.startswith(";;#ASMSTART"):
            n_stmt += 1
            stmt = n_stmt
            continue
        if s.startswith(";;#ASMEND"):
            stmt = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            out.append(dict(op="label", ops=[], asm=None, text=s, label=m.group(1)))
            continue
        s = s.split(";")[0].strip()
        if not s or s.startswith("."):
            continue
        op, _, rest = s.partition(" ")
        ops = [o.strip().split(" ")[0] for o in rest.split(",")] if rest else []
        out.append(dict(op=op, ops=ops, asm=stmt, text=s, label=None))
    return out


def defs_uses(ins):
    op, ops = ins["op"], ins["ops"]
    allr = lambda xs: sum((_regs(o) for o in xs), [])
    if op.startswith("v_mfma"):
        return _regs(ops[0]), allr(ops[1:4])
    if op.startswith(("buffer_load", "global_load", "scratch_load", "ds_read", "flat_load")):
        return _regs(ops[0]), allr(ops[1:])
    if op.startswith(("buffer_store", "global_store", "scratch_store", "ds_write", "flat_store", "buffer_atomic", "global_atomic", "ds_add", "ds_max")):
        return [], allr(ops)
    if op.startswith("v_accvgpr_write"):
        return _regs(ops[0]), allr(ops[1:])
    if op.startswith("v_cmp") and not op.startswith("v_cmpx"):
        return [], allr(ops)
    if op.startswith("v_") and ops:
        return _regs(ops[0]), allr(ops[1:])
    return [], []


def expand_loops(ins):
    """Walk order: every backward-branch loop body is walked once more right after itself (nested loops included, each once per level);
    of an if / else the `then` side is walked (an unconditional forward s_branch is followed)."""
    labels = {x["label"]: n for n, x in enumerate(ins) if x["label"]}

    def walk(start, end, done, depth):
        order, n = [], start
        while n <= end:
            order.append(n)
            x = ins[n]
            tgt = labels.get(x["ops"][0], None) if x["ops"] else None
            if re.match(r"s_cbranch_\w+|s_branch", x["op"]) and tgt is not None and tgt < n and tgt >= start and n not in done and depth < 3:
                order.extend(walk(tgt, n, done | {n}, depth + 1))      # the loop body once more (its own back edge not followed again)
            elif x["op"] == "s_branch" and tgt is not None and tgt > n:
                if tgt > end:
                    break
                n = tgt
                continue
            n += 1
        return order

    return walk(0, len(ins) - 1, frozenset(), 0)


def lint(text):
    ins = parse(text)
    order = expand_loops(ins)
    findings = []
    last_valu_write = {}    # reg -> issue position
    mfma_dest = {}          # reg -> (issue position, dest register tuple)
    store_data = {}         # reg -> issue position of the wide store reading it
    vm, lg = [], []         # in-flight memory operations, in order: (walk index, destination registers)
    pending = {}            # reg -> walk index of the load that will write it
    asm_written = {}        # statement id -> registers written so far inside it
    pos = 0

    vm_ids, lg_ids = set(), set()

    def retire(queue, keep):
        while len(queue) > keep:
            idx, dests = queue.pop(0)
            vm_ids.discard(idx)
            lg_ids.discard(idx)
            for r in dests:
                if pending.get(r) == idx:
                    del pending[r]

    def flag(rule, k, msg):
        findings.append(f"{rule} @{order[k]}: {ins[order[k]]['text']}  -- {msg}")

    for k, n in enumerate(order):
        x = ins[n]
        op = x["op"]
        if op == "label":
            continue
        if op == "s_waitcnt":
            for m in re.finditer(r"(vmcnt|lgkmcnt)\((\d+)\)", x["text"]):
                retire(vm if m.group(1) == "vmcnt" else lg, int(m.group(2)))
            if re.search(r"s_waitcnt\s+(0|0x0)\b", x["text"]):
                retire(vm, 0)
                retire(lg, 0)
            pos += 1
            continue
        if op == "s_nop":
            pos += int(x["ops"][0], 0) + 1
            continue
        if op == "s_barrier":
            pos += 1
            continue
        d, u = defs_uses(x)
        # R2 (a later load of the SAME in-order queue may overwrite a dead earlier one: the data lands in issue order)
        is_vm_load = op.startswith(("buffer_load", "global_load", "scratch_load", "flat_load"))
        is_lds_load = op.startswith("ds_read")
        for r in set(d + u):
            if r in pending:
                same_queue = r in d and r not in u and ((is_vm_load and pending[r] in vm_ids) or (is_lds_load and pending[r] in lg_ids))
                if not same_queue:
                    flag("R2", k, f"{r} is the destination of a load still in flight: {ins[order[pending[r]]]['text']}")
        # R3
        if x["asm"] is not None:
            if k == 0 or ins[order[k - 1]]["asm"] != x["asm"] or order[k - 1] != n - 1:
                asm_written[x["asm"]] = set()      # (first instruction of this execution of the statement: loops are walked twice)
            w = asm_written.setdefault(x["asm"], set())
            for r in u:
                if r in w:
                    flag("R3", k, f"{r} was written earlier in the same asm statement (missing early-clobber?)")
            w.update(d)
        is_mfma = op.startswith("v_mfma")
        if is_mfma:
            # R1
            for r in u:
                if r in last_valu_write and pos - last_valu_write[r] - 1 < 2:
                    flag("R1", k, f"{r} written by a VALU instruction {pos - last_valu_write[r] - 1} wait state(s) earlier (2 needed)")
            # R4 (SrcC partially overlapping an earlier MFMA's destination)
            srcc = tuple(_regs(x["ops"][3])) if len(x["ops"]) > 3 else ()
            for r in srcc:
                if r in mfma_dest and mfma_dest[r][1] != srcc and pos - mfma_dest[r][0] - 1 < MFMA_GUARD_STATES:
                    flag("R4", k, f"SrcC overlaps the destination {mfma_dest[r][1][0]}.. of an MFMA {pos - mfma_dest[r][0] - 1} states earlier")
                    break
        else:
            for r in set(d + u):
                if r in mfma_dest and pos - mfma_dest[r][0] - 1 < MFMA_GUARD_STATES:
                    flag("R4", k, f"{r} is the destination of an MFMA issued {pos - mfma_dest[r][0] - 1} wait states earlier")
        # R5 (VALU writes only: a load's data lands tens of cycles later)
        for r in (d if op.startswith("v_") else []):
            if r in store_data and pos - store_data[r] - 1 < 1:
                flag("R5", k, f"{r} is still being read by the wide store in front of it")
        # bookkeeping
        if op.startswith(("buffer_load", "global_load", "scratch_load", "flat_load")):
            vm.append((k, d))
            vm_ids.add(k)
            for r in d:
                pending[r] = k
        elif op.startswith(("ds_read",)):
            lg.append((k, d))
            lg_ids.add(k)
            for r in d:
                pending[r] = k
        elif op.startswith(("ds_write", "ds_add", "ds_max")):
            lg.append((k, []))
        elif op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            lg.append((k, []))
        elif op.startswith(("buffer_store", "global_store", "scratch_store", "flat_store", "buffer_atomic", "global_atomic")):
            vm.append((k, []))
            data = _regs(x["ops"][0]) if op.startswith("buffer_store") else (_regs(x["ops"][1]) if len(x["ops"]) > 1 else [])
            if len(data) > 2:
                for r in data:
                    store_data[r] = pos
        if is_mfma:
            dt = tuple(d)
            for r in d:
                mfma_dest[r] = (pos, dt)
                last_valu_write.pop(r, None)
        elif op.startswith("v_") and d:
            for r in d:
                last_valu_write[r] = pos
                mfma_dest.pop(r, None)
        pos += 1
    n_asm = len({x["asm"] for x in ins if x["asm"] is not None})
    return dict(findings=findings, instructions=sum(1 for x in ins if x["op"] != "label"), asm_statements=n_asm,
                mfma=sum(1 for x in ins if x["op"].startswith("v_mfma")))


def main():
    src, sub, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    text = Path(src).read_text() if src.endswith(".s") else compile_kernel(src, sub, extra)
    r = lint(text)
    for f in r["findings"]:
        print(f)
    print(f"{src} {sub} {' '.join(extra)}: {r['instructions']} instructions, {r['mfma']} MFMAs, {r['asm_statements']} asm statements, "
          f"{len(r['findings'])} finding(s)")
    sys.exit(1 if r["findings"] else 0)


if __name__ == "__main__":
    main()
