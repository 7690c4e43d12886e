"""The C-ABI library must load and export every symbol include/phonic_gpu.h declares (no compute calls: no GPU here).
Also checks the parameter descriptor tables against the reference constants (ids, ranges, defaults, scalings:
SURVEY.md Appendix A) through the descriptor entry points, which do not touch the device."""
import ctypes as C
import os
import re

import pytest

from phonic_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "phonic_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = C.CDLL(_capi.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_integration_md_declares_every_entry_point():
    """INTEGRATION.md §2 claims one `pub fn` per entry point of the header (several were missing): the Rust `extern "C"` block
    and the header must name exactly the same functions."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index('extern "C" {'):]
    block = block[:block.index("\n}\n```")]
    rust = set(re.findall(r"pub fn (pg_[a-z0-9_]+)\(", block))
    header = set(declared_symbols())
    assert rust == header, (sorted(header - rust), sorted(rust - header))


def test_effect_descriptors_match_reference_constants():
    lib = _capi.load()
    names = [lib.pg_effect_kind_name(k).decode() for k in range(10)]
    assert names == ["Gain", "Panning", "Filter", "Eq5", "Delay", "Reverb", "Chorus", "Compressor", "Gate", "Distortion"]
    # `fn weight` of each effect
    assert [lib.pg_effect_kind_weight(k) for k in range(10)] == [1, 1, 2, 3, 3, 5, 3, 4, 2, 1]
    from phonic_amd.graph import effect_parameters

    def P(kind):
        return {p["fourcc"]: p for p in effect_parameters(kind)}

    f = _capi.fourcc
    rv = P(_capi.FX_REVERB)  # reverb.rs:78-91
    assert rv[f("room")]["min"] == 0.0 and rv[f("room")]["max"] == 1.0 and abs(rv[f("room")]["default"] - 0.6) < 1e-7
    assert abs(rv[f("wet ")]["default"] - 0.35) < 1e-7
    ch = P(_capi.FX_CHORUS)  # chorus.rs:79-137
    assert ch[f("rate")]["scaling"] == 1 and ch[f("rate")]["scaling_args"][0] == 2.0 and ch[f("dlay")]["default"] == 12.0
    assert abs(ch[f("phas")]["max"] - 3.14159274) < 1e-6 and abs(ch[f("phas")]["default"] - 1.57079637) < 1e-6
    dl = P(_capi.FX_DELAY)  # delay.rs:124-177
    assert dl[f("dlay")]["min"] == 1.0 and dl[f("dlay")]["max"] == 4000.0 and dl[f("dlay")]["default"] == 375.0
    assert dl[f("cuto")]["default"] == 6000.0 and dl[f("lfos")]["n_values"] == 7 and dl[f("mode")]["n_values"] == 2
    cp = P(_capi.FX_COMPRESSOR)  # compressor.rs:44-91
    assert cp[f("thrs")]["default"] == -12.0 and cp[f("rato")]["default"] == 8.0 and abs(cp[f("look")]["default"] - 0.04) < 1e-7
    eq = effect_parameters(_capi.FX_EQ5)  # eq5.rs:38-150, parameters() order :246-264
    assert [p["fourcc"] for p in eq[:3]] == [f("gan1"), f("frq1"), f("bw_1")]
    assert [p["default"] for p in eq if p["name"].startswith("Frequency")] == [100.0, 1000.0, 4000.0, 8000.0, 12000.0]
    assert [p["max"] for p in eq if p["name"].startswith("Bandwidth")] == [1.0, 4.0, 4.0, 4.0, 1.0]
    ga = P(_capi.FX_GAIN)  # gain.rs:62-81
    assert ga[f("gain")]["scaling"] == 2 and ga[f("gain")]["scaling_args"] == (-60.0, 24.0) and abs(ga[f("gain")]["max"] - 15.848932) < 1e-5
    di = P(_capi.FX_DISTORTION)  # distortion.rs:209-228: default type = Diode (index 2)
    assert di[f("type")]["default"] == 2.0 and di[f("driv")]["max"] == 4.0
    gt = P(_capi.FX_GATE)
    assert gt[f("thrs")]["default"] == -30.0 and gt[f("rnge")]["default"] == -60.0
    fl = P(_capi.FX_FILTER)
    assert fl[f("cuto")]["default"] == 20000.0 and abs(fl[f("fltq")]["default"] - 0.707) < 1e-6 and fl[f("type")]["n_values"] == 4


def test_effect_create_validates_without_device():
    """pg_effect_create only builds the host description (no HIP call): errors follow the reference's taxonomy."""
    lib = _capi.load()
    init = _capi.make_init({"zzzz": 1.0})
    assert not lib.pg_effect_create(_capi.FX_REVERB, C.byref(init), 0)
    assert b"Unknown parameter" in lib.pg_last_error_message()
    init = _capi.make_init({"room": 2.0})  # out of range -> "Value out of bounds" (smoothed.rs:118-124)
    assert not lib.pg_effect_create(_capi.FX_REVERB, C.byref(init), 0)
    for shape in (5, 6):  # Random / Smooth Random LFO shapes: valid, their generator's state is an explicit input (default seed without one)
        h = lib.pg_effect_create(_capi.FX_DELAY, C.byref(_capi.make_init({"lfos": shape}, lfo_seed=(1, 2, 3, 4))), 0)
        assert h
        lib.pg_effect_destroy(h)
    assert not lib.pg_effect_create(_capi.FX_DELAY, C.byref(_capi.make_init({"lfos": 7})), 0)   # seven shapes: 0..6
    h = lib.pg_effect_create(_capi.FX_REVERB, C.byref(_capi.make_init({"room": 0.5})), 0)
    assert h
    assert lib.pg_effect_tail(h) > 0  # process_tail from the target values, host side
    assert lib.pg_effect_message_reset(h) == 0
    assert lib.pg_effect_process(h, None, 0, 0) == _capi.PG_ERR_STATE  # process before initialize
    lib.pg_effect_destroy(h)
    g = lib.pg_effect_create(_capi.FX_GAIN, C.byref(_capi.make_init()), 0)
    assert lib.pg_effect_message_reset(g) == _capi.PG_ERR_PARAMETER  # Gain has no message type
    lib.pg_effect_destroy(g)


def test_product_package_does_not_touch_the_oracle():
    """The shipped path must not import, link or load anything under oracle/."""
    pkg = os.path.join(ROOT, "phonic_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".h", ".hip", ".inl", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "libphonic_oracle" not in text and "po_graph" not in text and "import oracle" not in text, fn
    out = os.popen(f"readelf -d {_capi.LIB_PATH} 2>/dev/null").read()
    assert "oracle" not in out
