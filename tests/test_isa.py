"""What the shipped gfx950 code objects contain (CPU: llvm-objdump on lib/libmom4d.so, tools/isa_scan.py).

DESIGN.md section 0, hardware fact 3: packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in a kernel whose
other waves issue bf16 MFMAs produced wrong lanes about one launch in ten.  The build keeps the SLP vectoriser off
(csrc/Makefile: -fno-slp-vectorize), which is the only source of those forms in the HIP sources; nothing else would notice a
compiler upgrade or a flag change bringing them back, so this test reads the ISA that ships."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_scan  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(isa_scan.OBJDUMP), reason="llvm-objdump of the ROCm image is needed")


@pytest.fixture(scope="module")
def kernels():
    ks = {k: v for k, v in isa_scan.all_kernels().items() if not k.endswith(".kd")}
    assert len(ks) >= 40, "the library's code objects were not found"
    return ks


def test_no_packed_fp32_in_any_kernel_that_issues_mfma(kernels):
    mfma = {k for k, v in kernels.items() if any(i.startswith("v_mfma") for i in v)}
    assert any("deform_field_fwd_b3_kernel" in k for k in mfma) and any("deform_bwd_dx_kernel" in k for k in mfma)
    packed = isa_scan.packed_fp32(kernels)
    hit = {k: dict(c) for k, c in packed.items() if k in mfma}
    assert not hit, f"packed fp32 beside MFMAs (the hazard of DESIGN.md section 0): {hit}"


def test_no_packed_fp32_in_the_deformation_field_file(kernels):
    """Every kernel of csrc/deform_field.hip (they run concurrently with each other's MFMA waves on the second stream)."""
    src = open(os.path.join(ROOT, "iclr2025_3d-mom_amd", "csrc", "deform_field.hip")).read()
    names = set(re.findall(r"\b([a-z0-9_]+_kernel)\s*\(", src))
    assert "deform_field_fwd_b3_kernel" in names and "hexplane_bwd6_gather_kernel" in names
    packed = isa_scan.packed_fp32(kernels)
    hit = {k: dict(c) for k, c in packed.items() if any(n in k for n in names)}
    assert not hit, hit


def test_the_b3_forward_runs_on_the_bf16_pipe(kernels):
    b3 = [v for k, v in kernels.items() if "deform_field_fwd_b3_kernel" in k]
    assert len(b3) == 1
    n = sum(1 for i in b3[0] if i == "v_mfma_f32_32x32x16_bf16")
    assert n >= 96, n
    assert not any(i.startswith("scratch_") for i in b3[0]), "the fused field forward spills"


def test_compositing_kernels_do_not_spill(kernels):
    for pat in ("render_fwd_kernel", "render_bwd_kernelILb0"):
        ks = [v for k, v in kernels.items() if pat in k]
        assert ks
        for v in ks:
            assert not any(i.startswith("scratch_") for i in v), pat


def test_render_bwd_has_no_vector_write_of_exec_and_no_select_on_vcc_in_its_loop(kernels):
    """csrc/raster_render.hip, row_totals: the asm block of DPP adds begins with the two wait states of the VGPR hazard only; the
    five-wait-state DPP hazard is for VECTOR writes of EXEC (v_cmpx*), which the kernel must therefore not contain.  And the
    loop must stay free of v_cndmask_b32_e32 -- the VOP2 form reads VCC and costs 23.5 cycles (tools/probe/valu_rate.hip) --
    beyond the handful in the prologue (the __shfl_xor butterfly of wave_last)."""
    for k, v in kernels.items():
        if "render_bwd_kernel" in k:
            assert not any(i.startswith("v_cmpx") for i in v), k
            assert sum(1 for i in v if i == "v_cndmask_b32_e32") <= 10, k
