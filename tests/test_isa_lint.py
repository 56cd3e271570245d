"""Compiled-code check for a hazard no functional test catches reliably (tools/isa_lint.py).

Kernels that issue loads from inline assembly and wait for them with hand-counted `s_waitcnt`
(wino_conv_z_kernel's filter-operand ring, the fp16 filter gradient's transpose reads) rely on the
compiler never touching a destination register between the load and its wait.  Round 4's full-size
race test found hipcc copying the whole ring aside and back around the epilogue; a prefetch landing in
between was overwritten by the stale copy -- wrong results only when HBM was contended by another stream.
The lint reads hipcc's own assembly for the kernel sources and reports any such access; this test
compiles the two sources (device side only, no GPU needed) and requires a clean report."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "semi-supervised-adaptive-distillation_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))


HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc (ROCm) to produce the assembly")
@pytest.mark.parametrize("src,expect", [("conv3x3_winograd.hip", "wino_conv_z_kernel"),
                                        ("conv3x3_winograd24.hip", "wino24_conv_kernel"),
                                        ("conv3x3_f16.hip", "conv3x3_wgrad_f16_kernel"),
                                        ("conv3x3_split.hip", "conv3x3_split_kernel")])
def test_no_access_to_in_flight_asm_load_destinations(tmp_path, src, expect):
    import isa_lint
    out = str(tmp_path / (src + ".s"))
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                        "-I" + CSRC, "-fvisibility=hidden", "--cuda-device-only", "-S", "-o", out,
                        os.path.join(CSRC, "kernels", src)], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, "hipcc failed on %s:\n%s" % (src, r.stderr[-4000:])
    seen, findings = 0, []
    for name, lines in isa_lint.kernels(open(out).read()):
        ring, bad = isa_lint.lint(lines)
        if ring and expect in name:
            seen += 1
        findings += [(name, no, code) for no, code in bad]
        # round 6: a VALU-written scalar operand (spill restore, readfirstlane) inside 5 wait states of an asm VMEM
        findings += [(name, no, code, "s%s after %d wait states" % (reg, ws))
                     for no, code, reg, ws in isa_lint.lint_scalar_operands(lines)]
    assert seen >= 2, \
        "the lint found no asm-issued loads in %s: has the kernel changed shape?" % src
    assert not findings, findings[:10]


def test_lint_recognises_the_round3_shadow_copy():
    """The pattern it exists for, in miniature: an asm load, a compiler copy of its destination, the wait."""
    import isa_lint
    text = "\n".join([
        "_Zk:", "\t;;#ASMSTART", "\tbuffer_load_dwordx4 v[2:5], v150, s[12:15], s44 offen", "\t;;#ASMEND",
        "\tv_mov_b64_e32 v[34:35], v[2:3]",
        "\t;;#ASMSTART", "\ts_waitcnt vmcnt(7)", "\t;;#ASMEND",
        "\tv_mfma_f32_16x16x4_f32 v[82:85], v2, v111, v[82:85]",
        "\tv_cndmask_b32_e64 v2, 0, 1, s[56:57]", "\tv_cmp_ne_u32_e64 s[6:7], 1, v2",
        "\t.set _Zk.uses_flat_scratch, 0"])
    (name, lines), = list(isa_lint.kernels(text))
    ring, bad = isa_lint.lint(lines)
    assert ring == {2, 3, 4, 5}
    assert [code for _, code in bad] == ["v_mov_b64_e32 v[34:35], v[2:3]"]     # the MFMA and the later temp use are fine


def test_lint_releases_ownership_only_on_an_exact_vmcnt_zero():
    """`s_waitcnt vmcnt(0)` frees every in-flight destination; a wait on another count whose text
    merely contains it (vmcnt(01) never occurs, but lgkmcnt(0) / vmcnt(10) style neighbours do) must not."""
    import isa_lint
    def run(wait):
        text = "\n".join([
            "_Zk:", "\t;;#ASMSTART", "\tbuffer_load_dwordx4 v[2:5], v150, s[12:15], s44 offen", "\t;;#ASMEND",
            "\t;;#ASMSTART", "\t" + wait, "\t;;#ASMEND",
            "\tv_mov_b64_e32 v[34:35], v[2:3]", "\t.set _Zk.uses_flat_scratch, 0"])
        (_, lines), = list(isa_lint.kernels(text))
        return [code for _, code in isa_lint.lint(lines)[1]]
    assert run("s_waitcnt vmcnt(0)") == []
    assert run("s_waitcnt vmcnt(0) lgkmcnt(0)") == []
    assert run("s_waitcnt lgkmcnt(0)") == ["v_mov_b64_e32 v[34:35], v[2:3]"]
    assert run("s_waitcnt vmcnt(10)") == ["v_mov_b64_e32 v[34:35], v[2:3]"]


def test_lint_recognises_the_round6_spill_restore_hazard():
    """conv3x3_split.hip's first persistent version, in miniature: hipcc restores a spilled offset with v_readlane and the
    inline-asm load behind it reads the scalar register before the write has landed; with the s_nop 4 the statement now
    carries, the same sequence is clean."""
    import isa_lint
    def kernel(nop):
        return ["_Zk:", "\tv_readlane_b32 s15, v255, 25", "\t;;#ASMSTART"] + ([nop] if nop else []) + [
            "\tbuffer_load_dwordx4 v[10:13], v195, s[28:31], s15 offen", "\t;;#ASMEND",
            "\tbuffer_load_dwordx4 v[14:17], v196, s[28:31], s15 offen",        # hipcc's own load: its business
            "\t.set _Zk.uses_flat_scratch, 0"]
    (_, lines), = list(isa_lint.kernels("\n".join(kernel(None))))
    bad = isa_lint.lint_scalar_operands(lines)
    assert [(code.split()[0], reg, ws) for _, code, reg, ws in bad] == [("buffer_load_dwordx4", 15, 0)]
    (_, lines), = list(isa_lint.kernels("\n".join(kernel("\ts_nop 4"))))
    assert isa_lint.lint_scalar_operands(lines) == []
    (_, lines), = list(isa_lint.kernels("\n".join(kernel("\ts_nop 2"))))
    assert len(isa_lint.lint_scalar_operands(lines)) == 1
