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


def main():
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
