"""BASELINE.json config 1 — one preloaded mono WAV source, gain + pan only, offline WAV output — through the harness that mirrors the
reference's WavOutput pull loop (src/output/wav.rs:210-250; phonic_amd/offline.py). CPU part: WAV decode / encode known answers and
the pull loop over the oracle graph against a closed-form expectation. GPU part: the same files rendered by the HIP graph."""
import struct
import wave

import numpy as np
import pytest

import oracle
from phonic_amd import offline

SR = 44100


def cowbell_like(n=7375):
    """A mono 16-bit one-shot of the length of assets/cowbell.wav: two decaying partials (540 / 800 Hz)."""
    t = np.arange(n, dtype=np.float64) / SR
    x = 0.6 * np.exp(-t * 18.0) * (np.sin(2 * np.pi * 540.0 * t) + 0.7 * np.sin(2 * np.pi * 800.0 * t + 0.3))
    return np.clip(np.round(x * 32767.0), -32768, 32767).astype("<i2")


def write_pcm16(path, i16, channels=1, rate=SR):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(channels); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes(i16.tobytes())


def test_decode_known_answers(tmp_path):
    i16 = np.array([0, 1, -1, 16384, -16384, 32767, -32768, 12345], "<i2")
    write_pcm16(tmp_path / "a.wav", i16)
    pcm, ch, rate = offline.read_wav(tmp_path / "a.wav")
    assert (ch, rate) == (1, SR)
    assert np.array_equal(pcm[:-1], i16.astype(np.float32) / np.float32(32768.0)) and pcm[-1] == 0.0  # + the extra zero frame
    assert pcm[3] == 0.5 and pcm[6] == -1.0
    # 8-bit unsigned, 24-bit, 32-bit int, float32, an odd-sized chunk in front of the data chunk, WAVE_FORMAT_EXTENSIBLE
    def riff(tag, channels, bits, body, extra=b"", ext=False):
        fmt = struct.pack("<HHIIHH", 0xFFFE if ext else tag, channels, 48000, 48000 * channels * bits // 8, channels * bits // 8, bits)
        if ext:
            fmt += struct.pack("<HHI", 22, bits, 0) + struct.pack("<H", tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
        r = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + extra + b"data" + struct.pack("<I", len(body)) + body
        return b"RIFF" + struct.pack("<I", len(r)) + r
    x, ch, rate = offline.decode_wav_bytes(riff(1, 2, 8, bytes([128, 0, 255, 192])))
    assert (ch, rate) == (2, 48000) and np.array_equal(x, np.array([0.0, -1.0, 127 / 128, 0.5], np.float32))
    x, _, _ = offline.decode_wav_bytes(riff(1, 1, 24, bytes([0, 0, 0x40, 0, 0, 0xC0, 0xFF, 0xFF, 0x7F, 0, 0, 0x80])))
    assert np.array_equal(x, np.array([0.5, -0.5, 8388607 / 8388608, -1.0], np.float32))
    x, _, _ = offline.decode_wav_bytes(riff(1, 1, 32, np.array([1 << 30, -(1 << 31)], "<i4").tobytes(), extra=b"LIST" + struct.pack("<I", 3) + b"abc\x00"))
    assert np.array_equal(x, np.array([0.5, -1.0], np.float32))
    f = np.array([0.25, -0.75, 1.5], "<f4")
    x, _, _ = offline.decode_wav_bytes(riff(3, 1, 32, f.tobytes(), ext=True))
    assert np.array_equal(x, f)
    with pytest.raises(ValueError):
        offline.decode_wav_bytes(b"RIFX" + b"\0" * 40)
    with pytest.raises(ValueError):
        offline.decode_wav_bytes(riff(85, 1, 16, b"\0\0"))  # MP3-in-WAV: not PCM


def test_encode_float_wav_round_trip():
    s = np.linspace(-1.0, 1.0, 2 * 333, dtype=np.float32)
    data = offline.encode_wav_f32(s, 2, 48000)
    assert data[:4] == b"RIFF" and struct.unpack_from("<I", data, 4)[0] == len(data) - 8
    tag, ch, rate, brate, align, bits = struct.unpack_from("<HHIIHH", data, 20)
    assert (tag, ch, rate, brate, align, bits) == (3, 2, 48000, 48000 * 8, 8, 32)  # 32-bit float, as WavOutput writes (wav.rs:60-75)
    x, ch, rate = offline.decode_wav_bytes(data)
    assert (ch, rate) == (2, 48000) and np.array_equal(x, s)


def build_config1(g, pcm, channels, rate, volume=0.8, panning=-0.3):
    return g.add_voice(0, pcm, channels, rate, volume=volume, panning=panning)


def test_config1_pull_loop_on_the_oracle(tmp_path):
    """Mono 44.1 kHz one-shot into a 44.1 kHz stereo graph: resampler bypass (cubic.rs:53-58), mono -> stereo duplication
    (buffer.rs:209-217), constant gain then constant-power pan (smoothing.rs:60-122, utils.rs:56-62). The loop stops at the first
    call that writes nothing: the file holds whole 1024-frame blocks."""
    i16 = cowbell_like()
    write_pcm16(tmp_path / "cowbell_like.wav", i16)
    pcm, ch, rate = offline.read_wav(tmp_path / "cowbell_like.wav")
    g = oracle.OracleGraph(SR, 2, offline.BUFFER_SIZE_FRAMES)
    build_config1(g, pcm, ch, rate)
    frames = offline.render_to_wav(g, tmp_path / "out.wav", 2, SR)
    out, och, orate = offline.decode_wav_bytes(open(tmp_path / "out.wav", "rb").read())
    assert (och, orate) == (2, SR) and out.size == frames * 2
    assert frames % 1024 == 0 and frames >= i16.size and frames - i16.size < 2 * 1024
    x = i16.astype(np.float32) / np.float32(32768.0)
    n = np.float32((-0.3 + 1.0) / 2.0)
    pl = np.sqrt(np.float32(1.0) - n) / np.float32(1.0 / np.sqrt(2.0))
    pr = np.sqrt(n) / np.float32(1.0 / np.sqrt(2.0))
    want = np.zeros(frames * 2, np.float32)
    want[0 : 2 * i16.size : 2] = (x * np.float32(0.8)) * np.float32(pl)
    want[1 : 2 * i16.size : 2] = (x * np.float32(0.8)) * np.float32(pr)
    assert float(np.abs(out - want).max()) <= 2e-7
    # duration limit: whole seconds, checked before each pull (wav.rs:222-226)
    g2 = oracle.OracleGraph(SR, 2, offline.BUFFER_SIZE_FRAMES)
    g2.add_voice(0, pcm, ch, rate, has_repeat=1, repeat=(1 << 64) - 1)
    s = offline.render_blocks(g2, 2, SR, duration_seconds=1)
    assert s.size // 2 == -(-SR // 1024) * 1024  # the first block boundary at or past one second


@pytest.mark.gpu
@pytest.mark.parametrize("graph_rate", [44100, 48000])
def test_config1_gpu_matches_oracle(tmp_path, graph_rate):
    """Config 1 on the HIP graph against the oracle, file against file: at the file's own rate (bypass branch: bit-exact) and at
    48 kHz (cubic resampler 44.1 -> 48 kHz; schedule exact, samples within the f32 tolerance)."""
    from phonic_amd.graph import Graph

    i16 = cowbell_like()
    write_pcm16(tmp_path / "in.wav", i16)
    pcm, ch, rate = offline.read_wav(tmp_path / "in.wav")
    files = []
    for name, g in (("gpu", Graph(graph_rate, 2, offline.BUFFER_SIZE_FRAMES, 0)), ("cpu", oracle.OracleGraph(graph_rate, 2, offline.BUFFER_SIZE_FRAMES))):
        build_config1(g, pcm, ch, rate)
        offline.render_to_wav(g, tmp_path / f"{name}.wav", 2, graph_rate)
        files.append(offline.decode_wav_bytes(open(tmp_path / f"{name}.wav", "rb").read())[0])
    a, b = files
    assert a.size == b.size and a.size > 0 and np.abs(a).max() > 0.1
    if graph_rate == rate:
        assert np.array_equal(a, b)
    else:
        d = a.astype(np.float64) - b.astype(np.float64)
        assert float(np.sqrt(np.mean(d * d))) <= 1e-6 and float(np.abs(d).max()) <= 1e-5
