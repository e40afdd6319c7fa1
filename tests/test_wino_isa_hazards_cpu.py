"""CPU: the machine code of conv3x3_wino_bf16x9 (csrc/conv_wino_bf16.hip) is free of the hazards hipcc does not handle around inline asm
-- for the product build AND for builds whose vector-register assignment is deliberately different.

History: the kernel's split / transform arithmetic and its accumulator reset are inline asm under builtin MFMAs.  Round 5 found that any
edit which made the register allocator choose differently turned the results into inf / NaN and answered by pinning the K loop's machine
code with a digest.  Round 6 root-caused it (scripts/isa_hazards.py rule R1; confirmed on the MI355X, profiles/r06_wino_root_cause.txt):
the reset MFMAs read a zero operand the compiler had written with `v_mov` in the instruction directly in front of the asm statement --
0 wait states where the hardware needs 2 -- so the first MFMA multiplied the registers' previous contents; harmless while those happened to
be small integers, inf / NaN once they were never-written registers.  The K loop itself was never at fault.  The digest is gone; this test
checks the rules themselves (and tests/test_gpu_wino_permuted.py runs the parity tests on the permuted builds on the GPU)."""
import shutil
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "scripts"))
SRC = ROOT / "megapose6d_amd" / "csrc" / "conv_wino_bf16.hip"
# <DIAG 0, residual?, second output?, persistent?>: the persistent instances are what the pipeline launches, the others the MP_WINO_PERSIST=0 form
KERNELS = ("conv3x3_wino_bf16x9ILi0ELb0ELb0ELb1E", "conv3x3_wino_bf16x9ILi0ELb1ELb0ELb1E", "conv3x3_wino_bf16x9ILi0ELb1ELb1ELb1E",
           "conv3x3_wino_bf16x9ILi0ELb0ELb0ELb0E", "conv3x3_wino_bf16x9ILi0ELb1ELb0ELb0E")
needs_hipcc = pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="hipcc not available")


@needs_hipcc
@pytest.mark.parametrize("defs", [(), ("-DMP_WINO_PERMUTE=3",), ("-DMP_WINO_PERMUTE=8",)], ids=["product", "permuted3", "permuted8"])
def test_bf16_winograd_kernel_has_no_unpadded_hazard(defs):
    import isa_hazards

    kernels = isa_hazards.compile_all(SRC, defs)   # one compilation per build
    for kernel in KERNELS:
        text = next(v for k, v in kernels.items() if kernel in k)
        r = isa_hazards.lint(text)
        # 144 MFMAs in the K loop + 144 in the peeled last step + the 16 reset MFMAs (+ 16 more in the persistent form's store loop)
        assert r["mfma"] == (320 if kernel.endswith("Lb1E") else 304) and r["asm_statements"] >= 200, (kernel, r)
        assert r["findings"] == [], kernel + "\n" + "\n".join(r["findings"])


@needs_hipcc
def test_permuted_builds_really_have_another_k_loop_register_assignment():
    import isa_digest

    base = isa_digest.kloop_digest(str(SRC), KERNELS[1], 144)
    for n in (3, 8):
        d = isa_digest.kloop_digest(str(SRC), KERNELS[1], 144, extra=[f"-DMP_WINO_PERMUTE={n}"])
        assert d is not None and d["mfma"] == 144
        assert d["sha1"] != base["sha1"], "MP_WINO_PERMUTE no longer perturbs the K loop: pick another perturbation"


def test_the_lint_finds_the_round5_bug_and_accepts_the_fix():
    import isa_hazards

    old = """
	v_mov_b64_e32 v[134:135], s[10:11]
	v_mov_b64_e32 v[132:133], s[8:9]
	;;#ASMSTART
	v_mfma_f32_32x32x16_bf16 a[0:15], v[132:135], v[132:135], 0
	;;#ASMEND
	;;#ASMSTART
	v_mfma_f32_32x32x16_bf16 a[16:31], v[132:135], v[132:135], 0
	;;#ASMEND
"""
    f = isa_hazards.lint(old)["findings"]
    assert f and all(x.startswith("R1") for x in f), f
    fixed = old.replace("\tv_mfma", "\ts_nop 1\n\tv_mfma")
    assert isa_hazards.lint(fixed)["findings"] == []


def test_the_lint_rules_on_small_sequences():
    import isa_hazards

    # R2: a register whose load is still in flight is read by an asm statement
    f = isa_hazards.lint("""
	buffer_load_dwordx4 v[4:7], v1, s[4:7], 0 offen
	buffer_load_dwordx4 v[8:11], v1, s[4:7], 0 offen
	s_waitcnt vmcnt(1)
	;;#ASMSTART
	v_sub_f32 v20, v4, v8
	;;#ASMEND
""")["findings"]
    assert len(f) == 1 and f[0].startswith("R2") and "v8" in f[0], f
    # R3: missing early-clobber -- the second instruction of the statement reads what the first one wrote
    f = isa_hazards.lint("""
	;;#ASMSTART
	v_perm_b32 v10, v3, v2, s3
	v_perm_b32 v11, v5, v10, s3
	;;#ASMEND
""")["findings"]
    assert len(f) == 1 and f[0].startswith("R3"), f
    # R4: an MFMA's destination read by a VALU instruction straight after it
    f = isa_hazards.lint("""
	v_mfma_f32_32x32x16_bf16 v[0:15], v[20:23], v[24:27], v[0:15]
	v_add_f32 v40, v0, v1
""")["findings"]
    assert f and f[0].startswith("R4"), f
    # the accumulate chain itself (same registers as C and D) is fine, and so is a loop walked twice
    assert isa_hazards.lint("""
.LBB0_1:
	v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]
	v_mfma_f32_32x32x16_bf16 a[0:15], v[20:23], v[24:27], a[0:15]
	s_cbranch_scc1 .LBB0_1
""")["findings"] == []
