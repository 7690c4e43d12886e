"""The kernels' register, scratch and LDS budgets (phonic_amd/csrc/kernel_budget.json) against the built library: four workgroups per CU for the
staged and `mid` fast kernels need <= 128 VGPRs and <= 40 KB of LDS, and what a kernel may spill is written down per kernel with the measurement
that justifies it (the occupancy depended on compiler flags with nothing asserting the outcome). Reads the code object's
metadata; no GPU needed."""
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
    # the headline's kernel: four per CU, and at most the ONE spilled dword the budget file accounts for (an LDS address of the mid stage's set-up,
    # stored and reloaded once per block: 2 MB per 1024-voice block; the hand-over that causes it is + 2-3 % per real-time call, interleaved A/B)
    assert res["pg_stage_fused_kernel"]["vgpr"] <= 128 and res["pg_stage_fused_kernel"]["vgpr_spill"] <= 1 and res["pg_stage_fused_kernel"]["scratch"] <= 16
