"""The smoothers' grouped walk (phonic_amd/csrc/pg_dsp_dev.h sm_sequence: what every ramp path of the kernels lays its value sequences out with)
against a loop of sm_next, the reference's statement order (src/utils/smoothing.rs:21-28, 198-214, 360-382, 499-518): the same header compiled
for the host, 30 000 random states / targets / lengths of the three kinds, most of them with the ramp ending inside the call. Bit-equal
sequences and end states. (The device side of the same program runs on the GPU box: profiles/r05_smoother_walk.txt.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_grouped_walk_equals_the_per_value_walk(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this host")
    exe = str(tmp_path / "smooth.bin")
    src = os.path.join(ROOT, "tools", "exp_smooth", "smooth.hip")
    cmd = [hipcc, "-O3", "-ffp-contract=off", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "phonic_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-o", exe, src]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "--host-only"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 differ" in r.stdout
