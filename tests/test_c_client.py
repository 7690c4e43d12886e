"""The C ABI from plain C: tests/host/c_client.c is compiled as C99 (-Wall -Wextra -Werror -pedantic) against include/phonic_gpu.h and
linked to libphonic_gpu.so — the view a cgo / Rust `extern "C"` binding has. CPU: the header is valid C, the library
links, descriptor calls work without a device. GPU: the C program renders a small graph; the same graph through ctypes gives the
same numbers and the oracle agrees within the f32 tolerance."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from phonic_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "phonic_amd", "csrc")


def build_client(tmp_path):
    exe = str(tmp_path / "c_client")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "host", "c_client.c"), "-o", exe, "-L", CSRC, "-lphonic_gpu", "-lm",
           "-Wl,-rpath," + CSRC, "-Wl,--allow-shlib-undefined"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def run_client(exe, *args):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join(p for p in ("/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")) if p)
    return subprocess.run([exe, *args], check=True, capture_output=True, text=True, env=env, timeout=300).stdout


def test_header_is_c99_and_descriptors_work_without_a_device(tmp_path):
    if not os.path.exists(_capi.LIB_PATH):
        pytest.skip("libphonic_gpu.so not built")
    out = run_client(build_client(tmp_path), "describe").splitlines()
    assert out[0].split() == ["voice_defaults", "1.000", "0.000", "1.000", "0.050"]  # FilePlaybackOptions::default (file.rs:95-110)
    names = [l.split()[1] for l in out[1:11]]
    assert names == ["Gain", "Panning", "Filter", "Eq5", "Delay", "Reverb", "Chorus", "Compressor", "Gate", "Distortion"]
    assert "gain dcfm" in out[1] and "room" in out[6]
    assert out[11].split()[0] == "bad_kind_params" and int(out[11].split()[1]) <= 0


def _python_side(g, n_blocks):
    i = np.arange(4410, dtype=np.float64)
    l = (0.05 * np.sin(2.0 * 3.14159265358979323846 * 220.0 * i / 44100)).astype(np.float32)
    r = (0.05 * np.sin(2.0 * 3.14159265358979323846 * 222.2 * i / 44100 + 0.5)).astype(np.float32)
    pcm = np.concatenate([np.stack([l, r], axis=1).reshape(-1), np.zeros(2, np.float32)])
    m = g.add_mixer()
    fx_gain = g.add_effect(m, _capi.FX_GAIN, params={"gain": 0.5})
    g.add_effect(m, _capi.FX_REVERB, params={"room": 0.4}, reverb_seeds=(12345, 54321, [0.25 * k for k in range(16)]))
    g.add_effect(0, _capi.FX_EQ5)
    g.add_voice(m, pcm, 2, 44100, volume=0.8, panning=-0.25, has_repeat=1, repeat=_capi.PG_REPEAT_FOREVER)
    g.schedule_param(fx_gain, "gain", 0.25, 2 * 1024 + 100)
    rows = []
    for b in range(n_blocks):
        out = np.zeros(2048, np.float32)
        w = g.write(out, b * 1024)
        rows.append((w, float(out.astype(np.float64).sum()), float(np.abs(out.astype(np.float64)).sum())))
    return rows


@pytest.mark.gpu
def test_c_program_renders_the_same_graph_as_ctypes_and_oracle(tmp_path):
    from phonic_amd.graph import Graph

    n_blocks = 6
    lines = run_client(build_client(tmp_path), "render", str(n_blocks)).splitlines()
    c_rows = [(int(a), float(b), float(c)) for a, b, c in (l.split() for l in lines)]
    assert len(c_rows) == n_blocks and all(w == 2048 for w, _, _ in c_rows)
    py_rows = _python_side(Graph(48000, 2, 1024, 0), n_blocks)
    cpu_rows = _python_side(oracle.OracleGraph(48000, 2, 1024), n_blocks)
    for (wc, sc, ac), (wp, sp, ap), (wo, so, ao) in zip(c_rows, py_rows, cpu_rows):
        assert wc == wp == wo
        assert abs(sc - sp) <= 1e-7 * max(1.0, ap) and abs(ac - ap) <= 1e-7 * max(1.0, ap)   # same library, two bindings (printed with 10 digits)
        assert abs(ac - ao) <= 2048 * 1e-5 and abs(sc - so) <= 2048 * 1e-5                   # oracle: within the sample tolerance, summed
    assert c_rows[-1][2] > 1.0
