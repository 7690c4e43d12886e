"""Analytic known answers for the CPU oracle: the reference has no tests for its filters, delay
lines or effects, so the restatement is additionally checked against closed-form facts about the algorithms it restates."""
import ctypes as C

import numpy as np

import oracle
import workloads
from oracle import fp
from phonic_amd import _capi

SR = 48000


def sine(freq, n, sr=SR, amp=0.5):
    return (amp * np.sin(2 * np.pi * freq * np.arange(n) / sr)).astype(np.float32)


def rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))


def test_svf_lowpass_magnitude_at_cutoff_equals_q(oracle_lib):
    """Cytomic TPT SVF low-pass: |H(fc)| = Q (biquad.rs:174-183)."""
    for q in (0.5, 0.707, 2.0):
        x = sine(1000.0, 48000)
        y = x.copy()
        oracle_lib.po_biquad_run(0, SR, 1000.0, q, 0.0, fp(y), y.size)
        assert abs(rms(y[24000:]) / rms(x[24000:]) - q) < 0.01 * q


def test_svf_highpass_bandpass_notch(oracle_lib):
    x = sine(1000.0, 48000)
    y = x.copy()
    oracle_lib.po_biquad_run(2, SR, 1000.0, 2.0, 0.0, fp(y), y.size)  # Bandpass m1 = 1: |H(fc)| = Q
    assert abs(rms(y[24000:]) / rms(x[24000:]) - 2.0) < 0.03
    y = x.copy()
    oracle_lib.po_biquad_run(3, SR, 1000.0, 1.0, 0.0, fp(y), y.size)  # Notch at fc
    assert rms(y[24000:]) / rms(x[24000:]) < 1e-3
    lo = sine(50.0, 48000)
    y = lo.copy()
    oracle_lib.po_biquad_run(1, SR, 5000.0, 0.707, 0.0, fp(y), y.size)  # Highpass far below cutoff: -80 dB
    assert rms(y[24000:]) / rms(lo[24000:]) < 2e-4


def test_bell_and_shelf_gains(oracle_lib):
    """Bell: gain dB at fc; low shelf: gain dB at DC, 0 dB at high frequencies (biquad.rs:234-268)."""
    x = sine(1000.0, 48000, amp=0.1)
    y = x.copy()
    oracle_lib.po_biquad_run(6, SR, 1000.0, 1.0, 6.0, fp(y), y.size)
    assert abs(20 * np.log10(rms(y[24000:]) / rms(x[24000:])) - 6.0) < 0.05
    lo = sine(20.0, 96000, amp=0.1)
    y = lo.copy()
    oracle_lib.po_biquad_run(7, SR, 1000.0, 0.707, -9.0, fp(y), y.size)
    assert abs(20 * np.log10(rms(y[48000:]) / rms(lo[48000:])) + 9.0) < 0.1
    hi = sine(15000.0, 48000, amp=0.1)
    y = hi.copy()
    oracle_lib.po_biquad_run(7, SR, 200.0, 0.707, -9.0, fp(y), y.size)
    assert abs(20 * np.log10(rms(y[24000:]) / rms(hi[24000:]))) < 0.1


def test_dc_filter_blocks_dc_and_minus_3db_point(oracle_lib):
    """One-pole DC blocker R = 1 - 2*pi*hz/fs (dc.rs:54-61): removes DC, ~-3 dB at `hz`."""
    x = np.full(96000, 0.5, np.float32)
    y = x.copy()
    oracle_lib.po_dc_run(2, SR, fp(y), y.size)  # Fast: 20 Hz
    assert abs(y[-1]) < 1e-4
    s = sine(20.0, 192000)
    y = s.copy()
    oracle_lib.po_dc_run(2, SR, fp(y), y.size)
    assert abs(20 * np.log10(rms(y[96000:]) / rms(s[96000:])) + 3.0) < 0.2


def test_allpass_delay_line_is_allpass(oracle_lib):
    """Schroeder allpass g = 0.5 (delay.rs:314-350): unit energy gain for an impulse, first tap -0.5... structure."""
    n, delay = 4096, 37
    buf = np.zeros((n, 2), np.float64)
    buf[0] = [1.0, -2.0]
    oracle_lib.po_allpass_run(64, delay, buf.ctypes.data_as(C.POINTER(C.c_double)), n)
    # y[0] = 0.5*x, y[k*delay... ]: geometric tail; total energy = input energy
    assert abs(np.sum(buf[:, 0] ** 2) - 1.0) < 1e-9 and abs(np.sum(buf[:, 1] ** 2) - 4.0) < 1e-8
    assert buf[0, 0] == 0.5 and buf[delay, 0] == 0.75  # buf*0.5 ; then delayed(1.0) - 0.25


def test_interpolated_delay_impulse_weights(oracle_lib):
    """Fractional delay d: the impulse arrives at floor(d) and ceil(d) with linear weights (delay.rs:120-142)."""
    x = np.zeros(64, np.float32)
    x[0] = 1.0
    oracle_lib.po_interp_delay_run(64, 0.0, 10.25, fp(x), x.size)
    assert abs(x[10] - 0.75) < 1e-6 and abs(x[11] - 0.25) < 1e-6 and np.count_nonzero(x) == 2
    x = np.zeros(64, np.float32)
    x[0] = 1.0
    oracle_lib.po_interp_delay_run(64, 0.5, 8.0, fp(x), x.size)  # feedback 0.5: echoes every 8 frames, halving
    np.testing.assert_allclose(x[[8, 16, 24]], [1.0, 0.5, 0.25], atol=1e-6)


def test_lfo_sine_approx(oracle_lib):
    xs = np.linspace(-np.pi, np.pi, 101)
    err = [abs(oracle_lib.po_sine_approx(float(x)) - np.sin(x)) for x in xs]
    assert max(err) < 1.2e-3  # parabolic approximation with P = 0.225 (lfo.rs:9-19)


def test_effect_identities():
    """Default Gain/Panning/Distortion(mix=1, drive=0 Diode is NOT identity) / Eq5 (all gains 0 dB) leave the signal as is."""
    x = workloads.test_signal(2048, seed=3)
    for kind in (_capi.FX_GAIN, _capi.FX_PANNING, _capi.FX_EQ5):
        e = oracle.OracleEffect(kind)
        e.initialize(SR, 2, 4096)
        y = x.copy()
        e.process(y)
        assert np.array_equal(x, y), _capi.FX_NAMES[kind]
    e = oracle.OracleEffect(_capi.FX_PANNING, {"pan ": 1.0})  # hard right: l = 0, r = sqrt(2)
    e.initialize(SR, 2, 4096)
    y = x.copy()
    e.process(y)
    assert np.all(y[0::2] == 0.0)
    np.testing.assert_allclose(y[1::2], x[1::2] * np.float32(np.sqrt(2.0)), rtol=2e-7)


def test_reverb_wet_zero_and_tail():
    """wet = 0: the input (fed through sin/asin and three filters of a zero signal) comes back as the dry signal."""
    x = workloads.test_signal(4096, seed=5)
    e = oracle.OracleEffect(_capi.FX_REVERB, {"wet ": 0.0}, workloads.reverb_seeds(1))
    e.initialize(SR, 2, 4096)
    y = x.copy()
    e.process(y)
    np.testing.assert_allclose(y, x, atol=1e-6)
    e = oracle.OracleEffect(_capi.FX_REVERB, None, workloads.reverb_seeds(1))
    e.initialize(SR, 2, 4096)
    # process_tail (reverb.rs:449-467) for room 0.6
    size = 0.6**2 * 75 + 25
    fb = 1 - (1 - (0.82 - ((1 - 0.6) * 0.7 + size * 0.002))) ** 4
    assert abs(e.process_tail() - (int(79 * size) + int(int(79 * size) * np.log10(0.001) / np.log10(fb)))) <= 1


def test_compressor_limiter_holds_threshold():
    """Limiter (ratio 20 -> slope 1, look-ahead peak): steady sine above threshold is brought to the threshold level."""
    e = oracle.OracleEffect(_capi.FX_COMPRESSOR, {"thrs": -12.0, "rato": 20.0, "knee": 0.0, "gain": 0.0, "attk": 0.001, "look": 0.001})
    e.initialize(SR, 2, 4096)
    x = np.repeat(sine(1000.0, 4 * 4096, amp=0.9), 2)
    y = x.copy()
    for b in range(4):
        e.process(y[b * 8192:(b + 1) * 8192])
    peak = np.abs(y[-8192:]).max()
    assert abs(20 * np.log10(peak) + 12.0) < 0.5


def test_gate_closes_on_silence():
    """Below the threshold the gate gain settles at `range` dB (-60 dB: factor 0.001 or exactly 0, gate.rs:185-189)."""
    e = oracle.OracleEffect(_capi.FX_GATE, {"thrs": -30.0, "hold": 0.0, "rels": 0.01})
    e.initialize(SR, 2, 4096)
    x = np.full(2 * 4096, 1e-2, np.float32)  # -40 dB, below the -30 dB threshold
    e.process(x)
    assert np.abs(x[-100:]).max() <= 1e-2 * 0.00101
    loud = np.full(2 * 4096, 0.5, np.float32)  # above threshold: opens within the attack time
    e.process(loud)
    assert abs(loud[-1] - 0.5) < 1e-3
