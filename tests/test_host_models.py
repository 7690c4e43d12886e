"""CPU checks of the two exact closed forms the device code relies on.

* `f32_phase_advance` (pg_dsp_dev.h, host+device): compiled for the host with hipcc and compared with the serial f32 loop on
  200 000 random (phase, increment, steps) triples — bit for bit.
* the time-parallel resampler schedule (`sched_parallel`, pg_source_dev.h, device only): its integer model (closed form +
  rounding-table scan + restart at differing wrap decisions) is restated in numpy and compared with the serial f32 recurrence
  of cubic.rs:72-90, including chained 3000-block trajectories. The device implementation itself is checked on the GPU
  (tests/test_gpu_graph.py::test_resampler_schedule_bit_exact_over_many_blocks)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")


def test_f32_phase_advance_equals_serial_loop(tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "phase_check"
    subprocess.run([hipcc, "-O2", "-ffp-contract=off", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "phonic_amd", "csrc"),
                    os.path.join(HOST, "phase_advance_check.hip"), "-o", str(exe)], check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout


def test_device_log10f_restatement_equals_the_host_libm(tmp_path):
    """pg_log10f (the level detectors of the Compressor and the Gate): glibc's log10f restated operation by operation, so that threshold and knee
    decisions fall on the same frame on the device as in the reference on this platform — 4.2e7 arguments, bit for bit."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = tmp_path / "log10f_check"
    subprocess.run([hipcc, "-O2", "-ffp-contract=off", "-fno-builtin", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "phonic_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(HOST, "log10f_check.hip"), "-o", str(exe)], check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout


def test_device_expf_restatement_equals_the_host_libm(tmp_path):
    exe = tmp_path / "expf_check"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-ffp-contract=off", "-fno-builtin", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "phonic_amd", "csrc"),
                    "-I", os.path.join(ROOT, "include"), os.path.join(HOST, "expf_check.hip"), "-o", str(exe)], check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout


def test_resampler_schedule_integer_model_equals_serial_recurrence():
    sys.path.insert(0, HOST)
    import resampler_schedule_model as m

    rng = np.random.default_rng(5)
    fallbacks = 0
    for t in range(120):
        ratio = np.float32([44100 / 48000, 32000 / 48000, 0.5, 0.99999994, 0.75][t % 5]) if t < 40 else np.float32(rng.uniform(0.5, 1.0))
        if ratio >= 1.0:
            continue
        sp0 = np.float32(rng.integers(0, 2 * m.TWO24) / m.TWO24) if t % 3 else np.float32([0.0, 1.0, 0.99999994, 1.9999999][t % 4])
        oc, of, spn = m.serial(sp0, ratio, 1024)
        r = m.parallel(sp0, ratio, 1024)
        if r is None:
            fallbacks += 1
            continue
        cc, pf, spo = r
        assert np.array_equal(cc, oc)
        assert np.array_equal(pf.view(np.uint32), of.view(np.uint32))
        assert np.float32(spo).view(np.uint32) == spn.view(np.uint32)
    assert fallbacks < 20


def test_resampler_schedule_integer_model_for_ratios_above_one():
    """sched_parallel_up (pg_source_dev.h): ratio in [1, 4) — units of 2^-23 below 2, of 2^-22 above; the rounding of the push that crosses 2.0 / 4.0."""
    sys.path.insert(0, HOST)
    import resampler_schedule_model as m

    rng = np.random.default_rng(9)
    fallbacks = 0
    for t in range(360):
        lo, hi = [(1.0, 2.0), (2.0, 3.0), (3.0, 4.0)][t % 3]
        ratio = np.float32([44100 / 48000 * 1.5, 2.0, 3.9999998, 1.25, 88200 / 48000, 3.0, 2.5, 1.0, 1.9999999][t % 9]) if t < 45 else np.float32(rng.uniform(lo, hi))
        if not (1.0 <= ratio < 4.0):
            continue
        one = m.ONE if ratio < 2 else m.ONE // 2
        sp0 = np.float32(rng.integers(0, one) / one) if t % 4 else np.float32([0.0, 0.5, 0.75, 0.25][(t // 4) % 4])
        oc, of, spn = m.serial_up(sp0, ratio, 206)
        r = m.parallel_up(sp0, ratio, 206)
        if r is None:
            fallbacks += 1
            continue
        cc, pf, spo = r
        assert np.array_equal(cc, oc), (ratio, sp0)
        assert np.array_equal(pf.view(np.uint32), of.view(np.uint32)), (ratio, sp0)
        assert np.float32(spo).view(np.uint32) == spn.view(np.uint32)
    assert fallbacks < 12
