"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): the drop-in claim of INTEGRATION.md is
checked against the imported reference -- public signatures and the monkey-patch recipe."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(not Path("/root/reference/src/megapose/__init__.py").is_file(), reason="reference sources only exist in the build container")
def test_public_signatures_and_integration_patch_against_the_reference():
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "_ref_api_check.py")], capture_output=True, text=True, timeout=600)
    line = next((l for l in p.stdout.splitlines() if l.startswith("REF_API_JSON ")), None)
    assert line is not None, p.stdout[-2000:] + p.stderr[-4000:]
    problems = json.loads(line[len("REF_API_JSON "):])
    assert problems == [], "\n".join(problems)
