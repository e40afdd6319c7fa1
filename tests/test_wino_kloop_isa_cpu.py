"""CPU: the hand-scheduled K loop of conv3x3_wino_bf16x9 (csrc/conv_wino_bf16.hip) is pinned by its machine code.

Why a test on ISA: the loop's exact-piece split and transform arithmetic is inline asm placed slot by slot under the MFMAs; the schedule
was tuned AND validated on the MI355X for one vector-register assignment.  Round 5 measured what happens when an edit elsewhere in the
kernel (an L2 prefetch in the epilogue; a persistent outer loop) makes the allocator permute that assignment: the opcode sequence stays the
same, the kernel returns inf / NaN on every shape (profiles/r05_wino_persist_ab.txt).  There is no GPU in the build container, so this
test is the tripwire: it compiles the file (hipcc cross-compiles gfx950 here) and compares the loop -- 781 instructions, 144 MFMAs, scalar
register names normalised away -- with the digest of the kernel that passed the GPU parity tests.  If it fails after an intended change:
run `pytest -m gpu -k "winograd or backbone or exact_piece"` + scripts/microbench/native_wino_check on the GPU box, then update the digest.
"""
import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "scripts"))

# K loop of the kernel validated on the MI355X in round 4 (GPUTEST_r04: 243 passed) and again in round 5's final pass; identical, modulo
# scalar register names, before and after this round's telemetry gate
KLOOP_SHA1 = "56b3c3925908dc0b960549331961968581a4c1a2"
KLOOP_OPCODES_SHA1 = "0781da25187eea8d0675a1a64210c1b087452bda"


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="hipcc not available")
def test_bf16_winograd_k_loop_is_the_validated_machine_code():
    import isa_digest

    d = isa_digest.kloop_digest(str(ROOT / "megapose6d_amd" / "csrc" / "conv_wino_bf16.hip"), "conv3x3_wino_bf16x9ILi0E", 144)
    assert d is not None, "K loop (144 MFMAs) not found in conv3x3_wino_bf16x9<0>"
    assert d["instructions"] == 781, d
    assert d["opcode_sha1"] == KLOOP_OPCODES_SHA1, ("the K loop's instruction sequence changed", d)
    assert d["sha1"] == KLOOP_SHA1, ("the K loop's VECTOR REGISTER ASSIGNMENT changed (same opcodes): the inline-asm schedule is only validated "
                                     "for the pinned assignment -- re-run the GPU parity tests before updating the digest", d)
