"""CPU sanitizer builds (no GPU): the oracle's own test-suites under AddressSanitizer + UndefinedBehaviorSanitizer, and the HOST side of the
product library — graph construction, events and command lists, topology rebuilds, the control ring's drain, the sharded handle's routing —
compiled host-only against a stub HIP runtime (allocations from malloc, copies as memcpy, launches as no-ops) under the same sanitizers and
driven by a random plan generator."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gcc_runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_suites_under_asan_and_ubsan():
    """`make -C oracle asan`, then the KAT, analytic, graph-semantics and golden-vector suites in a child interpreter with the sanitizer runtime
    preloaded and the sanitized oracle in place of the -O2 one: any heap error or undefined behaviour in the restatement aborts the child."""
    asan = _gcc_runtime("libasan.so")
    if not asan:
        pytest.skip("no libasan for this gcc")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    lib = os.path.join(ROOT, "oracle", "_build", "libphonic_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan, PHONIC_ORACLE_LIB=lib, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    suites = ["tests/test_oracle_kats.py", "tests/test_oracle_analytic.py", "tests/test_oracle_graph.py", "tests/test_golden.py"]
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider"] + suites, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "passed" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-2000:]


def _build_host_fuzz(tmp_path, sanitize):
    """The library's translation units host-only + the stub HIP runtime + the plan driver, linked into one sanitized executable."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "phonic_amd", "csrc")
    san = ["-fsanitize=" + sanitize] + (["-fno-sanitize-recover=undefined"] if "undefined" in sanitize else [])
    flags = ["--cuda-host-only", "-std=c++17", "-O1", "-g", "-fPIC", "-ffp-contract=off", "-fno-omit-frame-pointer", "-DPG_FAST_WAVES=2", "-Wno-unused", "-Wno-unused-command-line-argument"] + san
    objs = []
    jobs = []
    kernel_tus = sorted(os.path.basename(f)[:-4] for f in __import__("glob").glob(os.path.join(csrc, "pg_k_*.hip")))   # one kernel per translation unit (host side: the stubs)
    assert len(kernel_tus) >= 7, kernel_tus
    for tu in ["pg_host", "pg_fxstate", "pg_effect", "pg_sharded", "pg_kernels"] + kernel_tus:
        o = str(tmp_path / (tu + ".o"))
        objs.append(o)
        jobs.append(subprocess.Popen([hipcc] + flags + ["-c", os.path.join(csrc, tu + ".hip"), "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    stub = str(tmp_path / "hip_stub.o")
    jobs.append(subprocess.Popen([hipcc] + flags + ["-c", os.path.join(ROOT, "tests", "host", "hip_stub.cpp"), "-x", "hip", "-o", stub], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for j in jobs:
        out, _ = j.communicate(timeout=900)
        assert j.returncode == 0, out[-3000:]
    # the module constructors name the (absent) device code objects of their translation units: define those symbols, empty
    undefined = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True).stdout.split()
    fat = tmp_path / "fatbins.cpp"
    fat.write_text("".join(f"extern \"C\" const char {s}[16] = {{0}};\n" for s in sorted(set(u for u in undefined if u.startswith("__hip_fatbin_")))))
    exe = str(tmp_path / "host_fuzz")
    link = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O1", "-g"] + san + [os.path.join(ROOT, "tests", "host", "host_fuzz.cpp"), str(fat), stub] + objs +
                          ["-o", exe, "-lpthread", "-ldl"], capture_output=True, text=True, timeout=600)
    assert link.returncode == 0, link.stderr[-3000:]
    return exe


def test_host_side_of_the_library_under_asan_and_ubsan(tmp_path):
    """The translation units of libphonic_gpu compiled HOST-ONLY (hipcc --cuda-host-only: no device code) with -fsanitize=address,undefined and
    linked against tests/host/hip_stub.cpp instead of the HIP runtime: the library's host logic — topology rebuilds, append-only id maps, the control
    ring's drain, event queues and per-piece command lists, the chunk / piece walk of long writes and its staging spans, host-fed rings, the
    sharded handle's routing and issuing threads, the effect handle — runs random plans through the C ABI (tests/host/host_fuzz.cpp) on the CPU.
    "Device" memory is malloc'ed, so every upload and ring write of the host code is bounds-checked; LeakSanitizer watches the handles' lifetimes."""
    exe = _build_host_fuzz(tmp_path, "address,undefined")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([exe, "16", "220"], capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0 and "ok" in run.stdout, (run.stdout[-1000:], run.stderr[-4000:])


def test_host_side_of_the_library_under_tsan(tmp_path):
    """The same build under ThreadSanitizer: the sharded handle issues every shard from a thread of its own (job hand-over by an atomic sequence
    number + condition variable, results read after the join), its id maps are append-only tables read without a lock, and per-device tables are
    built under a mutex by whichever thread gets there first."""
    exe = _build_host_fuzz(tmp_path, "thread")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    run = subprocess.run([exe, "10", "200"], capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0 and "ok" in run.stdout and "ThreadSanitizer" not in run.stderr, (run.stdout[-1000:], run.stderr[-4000:])
