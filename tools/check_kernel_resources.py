"""Build-time check of the kernels' register / scratch / LDS budgets: the occupancy the kernels were tuned for —
four workgroups of 256 lanes per CU for the staged and the `mid` fast kernels — depends on compiler flags and on an opaque lane index
(phonic_amd/csrc/Makefile); a compiler bump or an innocent edit that costs a few VGPRs would lose it silently. This script reads the
AMDGPU metadata of the gfx950 code object inside libphonic_gpu.so (llvm-objdump --offloading + llvm-readelf --notes) and compares every
kernel with phonic_amd/csrc/kernel_budget.json; __graft_entry__.build() and tests/test_kernel_budget.py run it.

    python tools/check_kernel_resources.py [--print]
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "phonic_amd", "csrc", "libphonic_gpu.so")
BUDGET = os.path.join(ROOT, "phonic_amd", "csrc", "kernel_budget.json")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib=LIB):
    """{kernel name (demangled prefix): {vgpr, agpr, sgpr, scratch, lds_static, vgpr_spill, sgpr_spill}} of the gfx950 code object in `lib`."""
    tmp = tempfile.mkdtemp(prefix="pgres_")
    try:
        work = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, work)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], check=True, capture_output=True, cwd=tmp)
        cos = [f for f in os.listdir(tmp) if "gfx950" in f]   # one code object per translation unit that holds kernels
        if not cos:
            raise RuntimeError("no gfx950 code object in " + lib)
        notes = "".join(subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, c)], check=True, capture_output=True, text=True).stdout for c in cos)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
        block = ".agpr_count:" + block
        f = lambda key: int(re.search(r"\." + key + r":\s*(\d+)", block).group(1))
        name = re.search(r"\.name:\s*(\S+)", block).group(1)
        m = re.match(r"_Z\d+([A-Za-z0-9_]+?)(?:8PgLaunch|P[A-Z].*|I.*)?$", name)
        short = re.match(r"_Z(\d+)", name)
        if short:
            n = int(short.group(1))
            short = name[2 + len(short.group(1)):2 + len(short.group(1)) + n]
        else:
            short = name
        out[short] = {"vgpr": f("vgpr_count"), "agpr": f("agpr_count"), "sgpr": f("sgpr_count"), "scratch": f("private_segment_fixed_size"),
                      "lds_static": f("group_segment_fixed_size"), "vgpr_spill": f("vgpr_spill_count"), "sgpr_spill": f("sgpr_spill_count")}
    return out


def dynamic_lds():
    """Dynamic LDS of the launches whose occupancy matters (pg_debug_lds_bytes): staged single launch at 1024 frames, the `mid` fast kernel with
    C3's effect kinds (Filter + Chorus)."""
    import ctypes as C

    sys.path.insert(0, ROOT)
    from phonic_amd import _capi

    lib = _capi.load()
    lib.pg_debug_lds_bytes.restype = C.c_size_t
    lib.pg_debug_lds_bytes.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
    return {"pg_stage_fused_kernel": lib.pg_debug_lds_bytes(0, 1024, 0), "pg_stage_fused_wide_kernel": lib.pg_debug_lds_bytes(2, 1024, 0), "pg_stage_fused_adapt_kernel": lib.pg_debug_lds_bytes(2, 1024, 0), "pg_unit_kernel_fast_mid": lib.pg_debug_lds_bytes(1, 1024, (1 << 2) | (1 << 6)),
            # the lean fast kernel with Gain / Panning only, and the wide one with C-like chains that hold no Reverb / Compressor (Filter, Eq5,
            # Delay, Distortion): the arenas that let three workgroups share a CU
            "pg_unit_kernel_fast": lib.pg_debug_lds_bytes(1, 1024, (1 << 0) | (1 << 1)),
            "pg_unit_kernel_fast_wide": lib.pg_debug_lds_bytes(1, 1024, (1 << 2) | (1 << 3) | (1 << 4) | (1 << 9))}


def check(verbose=False):
    res, budget = kernel_resources(), json.load(open(BUDGET))
    lds = dynamic_lds()
    problems = []
    for name, b in budget["kernels"].items():
        if name not in res:
            problems.append(f"{name}: kernel not found in the code object")
            continue
        r = res[name]
        total_lds = r["lds_static"] + lds.get(name, 0)
        if verbose:
            print(f"{name:32s} vgpr {r['vgpr']:3d} (+{r['agpr']} agpr)  scratch {r['scratch']:4d} B  vgpr spills {r['vgpr_spill']}  lds {total_lds}")
        if r["vgpr"] + r["agpr"] > b["max_vgpr"]:
            problems.append(f"{name}: {r['vgpr']} + {r['agpr']} VGPRs > {b['max_vgpr']} ({b['why']})")
        if r["scratch"] > b["max_scratch"]:
            problems.append(f"{name}: {r['scratch']} bytes of scratch per lane > {b['max_scratch']}")
        if r["vgpr_spill"] > b.get("max_vgpr_spill", 0):
            problems.append(f"{name}: {r['vgpr_spill']} spilled VGPRs")
        if "max_lds" in b and total_lds > b["max_lds"]:
            problems.append(f"{name}: {total_lds} bytes of LDS per workgroup > {b['max_lds']} ({b['why']})")
    return problems


if __name__ == "__main__":
    p = check(verbose="--print" in sys.argv)
    for line in p:
        print("BUDGET EXCEEDED:", line)
    sys.exit(1 if p else 0)
