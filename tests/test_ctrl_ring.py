"""CPU tests of the control path primitives (phonic_amd/csrc/pg_ctrl.h): the lock-free multi-producer ring every pg_graph_schedule_* /
pg_graph_set_voice_* call pushes into, and the append-only id tables those calls read — built with ThreadSanitizer and hammered by
producer threads while a consumer drains (the reference: handles push MixerMessages from any thread, the audio thread pops them in
process_messages, src/source/mixed.rs:113-194,294-499)."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_control_ring_under_sanitizers(tmp_path, sanitizer):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "ctrl_stress")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-pthread", os.path.join(ROOT, "tests", "host", "ctrl_stress.cpp"), "-o", exe],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, "4", "100000"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in run.stderr and "ERROR: AddressSanitizer" not in run.stderr, run.stderr[-3000:]
    d = json.loads(run.stdout.strip().splitlines()[-1])
    assert d["errors"] == 0 and d["messages"] == d["expected"] == 400000
