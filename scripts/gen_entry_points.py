#!/usr/bin/env python3
"""Regenerate the entry-point table of INTEGRATION.md from include/mp_engine.h.

The table sits between the two marker lines; everything else in INTEGRATION.md is hand-written.  `python scripts/gen_entry_points.py`
rewrites the block in place, `--check` exits 1 if the block on disk differs from what the header gives (what
tests/test_host_cpu.py::test_integration_md_lists_exactly_the_exported_entry_points runs, together with a comparison against the
symbols the shared library really exports).

Per entry point: the name, the header line of its declaration, and the first sentence of the comment block in front of it (for the
functions that share a comment with their neighbour: the section banner they sit under) -- which is where the header cites the
reference interface (file:line under /root/reference/src/megapose/) each one replaces.
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "mp_engine.h"
DEBUG_HEADER = ROOT / "include" / "mp_engine_debug.h"   # measurement / telemetry entry points: exported, not part of the product boundary
DOC = ROOT / "INTEGRATION.md"
BEGIN = "<!-- BEGIN ENTRY POINTS (generated from include/mp_engine.h by scripts/gen_entry_points.py -- do not edit by hand) -->"
END = "<!-- END ENTRY POINTS -->"

_DECL = re.compile(r"^(?:const\s+)?(?:[A-Za-z_][A-Za-z0-9_]*\s+)*[A-Za-z_][A-Za-z0-9_]*\s*\*?\s*(mp_[A-Za-z0-9_]+)\s*\(")


def _clean(comment: str) -> str:
    text = re.sub(r"/\*+|\*+/", " ", comment)
    text = re.sub(r"^\s*\*", " ", text, flags=re.M)
    text = re.sub(r"-{4,}", " ", text)
    text = re.sub(r"\s+", " ", text).strip()
    return text


def _first_sentence(text: str, limit: int = 240) -> str:
    # sentence end = ". " / ": " / "; " followed by a capital or the end -- "py:217" or "e.g. x" do not end one
    m = re.search(r"^(.+?[.;])(?=\s+[A-Z(`]|\s*$)", text)
    s = (m.group(1) if m else text).strip()
    if len(s) > limit:
        s = s[: limit - 1].rstrip() + "…"
    return s.replace("|", "/")


def parse_header(path: Path = HEADER):
    """-> list of (name, line number, summary).  The header is a sequence of comments and declarations; a run of one-line comments
    framed by rule lines (/* ----- */) is a section banner, any other comment belongs to the declaration that follows it directly."""
    lines = path.read_text().splitlines()
    items = []            # ("comment", text, is_rule, has_rule_chars) | ("code", line, no)
    k = 0
    while k < len(lines):
        st = lines[k].strip()
        if st.startswith("/*"):
            buf = [lines[k]]
            while "*/" not in lines[k]:
                k += 1
                buf.append(lines[k])
            raw = "\n".join(buf)
            items.append(("comment", _clean(raw), _clean(raw) == "" and "----" in raw, k + 1))
        elif st:
            items.append(("code", st, False, k + 1))
        k += 1
    out, banner, in_banner, pending, depth = [], "", False, "", 0
    for kind, text, is_rule, no in items:
        if kind == "comment":
            if is_rule:
                in_banner = not in_banner
                if in_banner:
                    banner = ""
            elif in_banner:
                banner = (banner + " " + text).strip()
            else:
                pending = text
            continue
        if depth == 0:
            m = _DECL.match(text)
            if m and not text.startswith(("typedef", "#", "return", "extern")):
                out.append((m.group(1), no, _first_sentence(pending or banner)))
        if not text.startswith('extern "C"') and text != "}":   # (the extern "C" { ... } wrapper is not a body)
            depth += text.count("{") - text.count("}")
        if text.endswith((";", "{", "}")) or text.startswith("#"):
            pending = ""   # a comment describes only the declaration that follows it directly
    seen, uniq = set(), []
    for name, no, summary in out:
        if name not in seen:
            seen.add(name)
            uniq.append((name, no, summary))
    return uniq


def render_block(entries, debug_entries=None) -> str:
    if debug_entries is None:
        debug_entries = parse_header(DEBUG_HEADER) if DEBUG_HEADER.is_file() else []
    rows = [BEGIN, "", f"{len(entries)} entry points of the product boundary (`extern \"C\"`, `include/mp_engine.h`; line = where it is declared):", "",
            "| entry point | header line | what the header says it is / replaces (first sentence) |", "|---|---|---|"]
    for name, no, summary in entries:
        rows.append(f"| `{name}` | {no} | {summary} |")
    rows += ["", f"{len(debug_entries)} measurement / telemetry entry points (`include/mp_engine_debug.h`: exported by the same library, called by bench.py "
                 "and scripts/ only -- not what an integrator binds):", "",
             "| entry point | header line | what it reports (first sentence) |", "|---|---|---|"]
    for name, no, summary in debug_entries:
        rows.append(f"| `{name}` | {no} | {summary} |")
    rows += ["", END]
    return "\n".join(rows)


def current_block(text: str):
    i, j = text.find(BEGIN), text.find(END)
    if i < 0 or j < 0:
        return None
    return text[i: j + len(END)]


def main(argv) -> int:
    entries = parse_header()
    block = render_block(entries)
    text = DOC.read_text()
    cur = current_block(text)
    if "--check" in argv:
        if cur != block:
            print("INTEGRATION.md entry-point table is stale: run python scripts/gen_entry_points.py", file=sys.stderr)
            return 1
        return 0
    if cur is None:
        print("markers not found in INTEGRATION.md", file=sys.stderr)
        return 2
    DOC.write_text(text.replace(cur, block))
    print(f"{len(entries)} entry points written")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
