"""Offline WAV harness: the reference's `WavOutput` pull loop over a mixer graph (src/output/wav.rs:25,60-123,210-250) plus the
small amount of WAV I/O around it — what BASELINE.json's config 1 ("1 preloaded WAV source, gain+pan only, offline WAV output")
and SURVEY.md §8(f) rank 4 ask for. Host plumbing only: every sample is rendered by the graph handed in (the GPU graph in the
product; the tests pass the CPU oracle's graph through the same functions and compare the files).

  pcm, channels, rate = read_wav("cowbell.wav")            # decoded like the reference's preloaded file buffer
  g = phonic_amd.Graph(44100, 2, 1024)
  g.add_voice(0, pcm, channels, rate, volume=0.8, panning=-0.3)
  render_to_wav(g, "out.wav")                              # 1024-frame blocks until the graph stops producing output
"""
import struct

import numpy as np

BUFFER_SIZE_FRAMES = 1024  # src/output/wav.rs:25
DEFAULT_SAMPLE_RATE = 44100  # src/output/wav.rs:21
DEFAULT_CHANNEL_COUNT = 2  # src/output/wav.rs:22

_WAVE_FORMAT_PCM = 1
_WAVE_FORMAT_IEEE_FLOAT = 3
_WAVE_FORMAT_EXTENSIBLE = 0xFFFE


def _chunks(data):
    """(id, payload) of every RIFF sub-chunk of a WAVE file (word-aligned; a truncated last chunk is clipped to the file)."""
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos = 12
    while pos + 8 <= len(data):
        cid, size = data[pos : pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        yield cid, data[pos + 8 : pos + 8 + size]
        pos += 8 + size + (size & 1)


def decode_wav_bytes(data):
    """Interleaved f32 samples, channel count and sample rate of a PCM (8/16/24/32 bit) or IEEE-float (32/64 bit) WAVE file.
    Integer samples scale by 2^-(bits-1) (unsigned 8 bit: (x - 128) / 128) — the conversion the reference's decoder applies before
    samples reach the path (symphonia's `f32: FromSample<i16>` etc.; decoding itself is upstream of the hot path)."""
    fmt = None
    payload = None
    for cid, body in _chunks(data):
        if cid == b"fmt " and fmt is None:
            if len(body) < 16:
                raise ValueError("short fmt chunk")
            tag, channels, rate, _brate, align, bits = struct.unpack_from("<HHIIHH", body, 0)
            if tag == _WAVE_FORMAT_EXTENSIBLE and len(body) >= 26:
                tag = struct.unpack_from("<H", body, 24)[0]  # first two bytes of the sub-format GUID
            fmt = (tag, channels, rate, align, bits)
        elif cid == b"data" and payload is None:
            payload = body
    if fmt is None or payload is None:
        raise ValueError("missing fmt or data chunk")
    tag, channels, rate, align, bits = fmt
    if channels < 1 or rate < 1:
        raise ValueError("invalid channel count or sample rate")
    width = bits // 8
    n = len(payload) // width if width else 0
    n -= n % channels
    raw = payload[: n * width]
    if tag == _WAVE_FORMAT_PCM:
        if bits == 8:
            x = (np.frombuffer(raw, np.uint8).astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
        elif bits == 16:
            x = np.frombuffer(raw, "<i2").astype(np.float32) / np.float32(32768.0)
        elif bits == 24:
            b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float32) / np.float32(8388608.0)
        elif bits == 32:
            x = (np.frombuffer(raw, "<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise ValueError(f"unsupported PCM width: {bits} bits")
    elif tag == _WAVE_FORMAT_IEEE_FLOAT:
        if bits == 32:
            x = np.frombuffer(raw, "<f4").astype(np.float32)
        elif bits == 64:
            x = np.frombuffer(raw, "<f8").astype(np.float32)
        else:
            raise ValueError(f"unsupported float width: {bits} bits")
    else:
        raise ValueError(f"unsupported WAVE format tag: {tag}")
    return np.ascontiguousarray(x, dtype=np.float32), int(channels), int(rate)


def read_wav(path):
    """(pcm, channels, rate) ready for `add_voice`: the decoded interleaved buffer plus the one extra zero frame the reference's
    preloaded buffer carries at its end (src/source/file/buffer.rs:103-104)."""
    with open(path, "rb") as f:
        x, channels, rate = decode_wav_bytes(f.read())
    return np.concatenate([x, np.zeros(channels, np.float32)]), channels, rate


def encode_wav_f32(samples, channels, rate):
    """Bytes of a 32-bit IEEE-float WAVE file ("Wav files contents are always saved as 32bit floats", src/output/wav.rs:60-75)."""
    samples = np.ascontiguousarray(samples, dtype="<f4")
    body = samples.tobytes()
    fmt = struct.pack("<HHIIHH", _WAVE_FORMAT_IEEE_FLOAT, channels, rate, rate * channels * 4, channels * 4, 32)
    fact = struct.pack("<I", samples.size // max(channels, 1))
    riff = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact + b"data" + struct.pack("<I", len(body)) + body
    return b"RIFF" + struct.pack("<I", len(riff)) + riff


def render_blocks(graph, channels=DEFAULT_CHANNEL_COUNT, sample_rate=DEFAULT_SAMPLE_RATE, duration_seconds=None, block_frames=BUFFER_SIZE_FRAMES):
    """`WavStream::process` in a loop (src/output/wav.rs:210-250): pull `block_frames` frames per call at
    `pos_in_frames = playback_pos / channels`; stop once the whole-second position reaches `duration_seconds` or the source writes
    nothing; keep the `written` samples of each call; advance the position by the FULL buffer length (wav.rs:247). The global volume
    of the stream stays at its default 1.0, for which `apply_smoothed_gain` leaves the buffer alone (smoothing.rs:60-71).
    Returns the interleaved f32 samples."""
    buf = np.zeros(block_frames * channels, np.float32)
    out = []
    playback_pos = 0  # in samples, like the reference
    while True:
        pos_in_frames = playback_pos // channels
        if duration_seconds is not None and pos_in_frames // sample_rate >= duration_seconds:
            break
        written = int(graph.write(buf, pos_in_frames))
        if written == 0:
            break
        out.append(buf[:written].copy())
        playback_pos += buf.size
    return np.concatenate(out) if out else np.zeros(0, np.float32)


def render_to_wav(graph, path, channels=DEFAULT_CHANNEL_COUNT, sample_rate=DEFAULT_SAMPLE_RATE, duration_seconds=None):
    """Render `graph` through the WavOutput pull loop into a 32-bit float WAVE file; returns the number of frames written.
    `sample_rate` / `channels` must be the graph's own (WavOutput hands its specs to the player, src/player.rs)."""
    samples = render_blocks(graph, channels, sample_rate, duration_seconds)
    with open(path, "wb") as f:
        f.write(encode_wav_f32(samples, channels, sample_rate))
    return samples.size // channels
