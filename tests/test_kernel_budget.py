"""The kernels' register, scratch and LDS budgets (phonic_amd/csrc/kernel_budget.json) against the built library: four workgroups per CU for the
staged and `mid` fast kernels need <= 128 VGPRs and <= 40 KB of LDS, nothing may spill VGPRs (the occupancy depended on
compiler flags with nothing asserting the outcome). Reads the code object's metadata; no GPU needed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_kernels_stay_inside_their_resource_budgets():
    import check_kernel_resources as ckr

    if not os.path.exists(ckr.LLVM):
        import pytest

        pytest.skip("no ROCm LLVM tools on this host")
    res = ckr.kernel_resources()
    assert {"pg_stage_fused_kernel", "pg_unit_kernel_fast_mid", "pg_unit_kernel", "pg_mix_kernel"} <= set(res)
    assert ckr.check() == []
    assert res["pg_stage_fused_kernel"]["vgpr"] <= 128 and res["pg_stage_fused_kernel"]["vgpr_spill"] == 0
