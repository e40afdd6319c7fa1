#!/usr/bin/env python
"""Per-kernel digest of the gfx950 code inside a .hip translation unit (no GPU needed).

    python scripts/isa_digest.py megapose6d_amd/csrc/conv.hip [-D...]  > before.json
    ... edit ...
    python scripts/isa_digest.py megapose6d_amd/csrc/conv.hip          > after.json
    python scripts/isa_digest.py --diff before.json after.json

Compiles the file to device assembly (`hipcc -S --cuda-device-only`), splits it into kernels, strips what does not change the
executed code (comments, local label numbers, symbol names) and prints {demangled kernel: {sha1, instructions, vgprs, sgprs,
scratch, lds}}.  Used to prove that adding a template parameter / a new instantiation leaves the existing kernels' machine code
untouched when no GPU is at hand to re-measure them."""
import hashlib
import json
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{ROOT / 'include'}", "-Wno-unused-function"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return out.stdout.splitlines()


def digest(src: str, extra):
    asm = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", "-o", "-", src], capture_output=True,
                         text=True, check=True).stdout
    kernels = {}
    cur, body = None, []
    meta = {}
    for line in asm.splitlines():
        m = re.match(r"^(\w+):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            cur, body = m.group(1), []
            kernels[cur] = body
            continue
        if cur is None:
            continue
        if line.strip().startswith(".amdhsa_kernel"):
            cur = None
            continue
        s = line.split(";")[0].strip()
        if not s or s.startswith("."):
            if re.match(r"^\.L\w+:", s):
                body.append("L:")
            continue
        s = re.sub(r"\.L\w+", ".L", s)          # local labels are numbered per translation unit
        s = re.sub(r"_Z\w+", "SYM", s)          # symbol names (template arguments are part of them)
        body.append(s)
    for m in re.finditer(r"\.amdhsa_kernel (\w+)(.*?)\.end_amdhsa_kernel", asm, re.S):
        d = {}
        for key, tag in (("vgprs", "next_free_vgpr"), ("sgprs", "next_free_sgpr"), ("scratch", "private_segment_fixed_size"),
                         ("lds", "group_segment_fixed_size"), ("accum_offset", "accum_offset")):
            mm = re.search(rf"\.amdhsa_{tag} (\d+)", m.group(2))
            d[key] = int(mm.group(1)) if mm else None
        meta[m.group(1)] = d
    names = [k for k in kernels if k in meta]
    pretty = demangle(names)
    out = {}
    for k, p in zip(names, pretty):
        text = "\n".join(kernels[k])
        out[p] = {"sha1": hashlib.sha1(text.encode()).hexdigest(), "instructions": sum(1 for l in kernels[k] if l != "L:"), **meta[k]}
    return out


def kloop_digest(src: str, kernel_substr: str, n_mfma: int, extra=()):
    """The innermost loop of the kernel whose mangled name contains `kernel_substr` that holds exactly `n_mfma` v_mfma instructions: its
    instruction list with scalar register NAMES normalised away (they do not affect timing or the VALU -> MFMA distances), sha1 + counts.
    Used to pin the hand-scheduled K loop of conv3x3_wino_bf16x9: its split / transform arithmetic is inline asm, the schedule was tuned and
    validated for ONE vector-register assignment, and an edit elsewhere in the kernel can permute that assignment without changing the
    opcode sequence -- measured in round 5 to turn the results into inf / NaN (profiles/r05_wino_persist_ab.txt)."""
    asm = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "-S", "--cuda-device-only", "-o", "-", src], capture_output=True,
                         text=True, check=True).stdout
    parts = re.split(r"\n(_Z\w+):[^\n]*\n", asm)
    for i in range(1, len(parts), 2):
        if kernel_substr not in parts[i]:
            continue
        lines = parts[i + 1].split("s_endpgm")[0].splitlines()
        labels = {m.group(1): n for n, l in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        best = None
        for n, l in enumerate(lines):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < n:
                a = labels[m.group(1)]
                if sum("v_mfma" in x for x in lines[a:n]) == n_mfma and (best is None or n - a < best[1] - best[0]):
                    best = (a, n)
        if best is None:
            continue
        body = [re.sub(r"\.LBB\d+_\d+", "L", x.split(";")[0].strip()) for x in lines[best[0]: best[1] + 1]]
        body = [re.sub(r"\bs\d+\b|s\[\d+:\d+\]", "S", x) for x in body if x]
        ops = [x.split()[0] for x in body]
        return {"kernel": parts[i], "instructions": len(body), "mfma": n_mfma, "sha1": hashlib.sha1("\n".join(body).encode()).hexdigest(),
                "opcode_sha1": hashlib.sha1("\n".join(ops).encode()).hexdigest()}
    return None


def main():
    if sys.argv[1] == "--kloop":   # --kloop file.hip kernel-substring n_mfma
        print(json.dumps(kloop_digest(sys.argv[2], sys.argv[3], int(sys.argv[4])), indent=1))
        return
    if sys.argv[1] == "--diff":
        a, b = (json.loads(Path(p).read_text()) for p in sys.argv[2:4])
        rc = 0
        for k in sorted(set(a) | set(b)):
            # a new trailing template argument shows up in the name: match "f<1, 2>" with "f<1, 2, false>"
            kb = k if k in b else next((n for n in b if n.replace(", false>", ">") == k), None)
            if k not in a:
                continue
            if kb is None:
                print(f"GONE      {k}")
                rc = 1
            elif a[k] != b[kb]:
                print(f"CHANGED   {k}: {a[k]} -> {b[kb]}")
                rc = 1
            else:
                print(f"identical {k}")
        for k in sorted(b):
            if k not in a and k.replace(", false>", ">") not in a:
                print(f"new       {k}: {b[k]}")
        sys.exit(rc)
    print(json.dumps(digest(sys.argv[1], sys.argv[2:]), indent=1))


if __name__ == "__main__":
    main()
