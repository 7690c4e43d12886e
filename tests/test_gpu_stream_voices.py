"""Host-fed sources (pg_graph_add_stream_voice / pg_graph_feed_voice): any `dyn Source` the host pulls itself — a synth, a streamed file
(MixerMessage::AddSource, src/source/mixed.rs:117-123; src/source/synth/common.rs:194-263) — inside a GPU mixer. The device reads the
fed ring where a file voice reads its preloaded buffer, behind the same adapters (ResampledSource, mono -> stereo, volume, panning), so
a fed voice must equal a preloaded voice holding the same PCM BIT FOR BIT; the preloaded path is the one held against the oracle."""
import numpy as np
import pytest

import workloads
from phonic_amd import _capi

pytestmark = pytest.mark.gpu
SR = 48000


def graph(max_frames=1024):
    from phonic_amd.graph import Graph

    return Graph(SR, 2, max_frames, 0)


def pcm_for(i, rate, seconds, channels):
    return workloads.tone_buffer(i, rate, seconds, channels=channels)      # incl. the decoder's extra zero frame


@pytest.mark.parametrize("channels,rate", [(2, 48000), (1, 48000), (2, 44100), (1, 32000), (2, 96000)])
@pytest.mark.parametrize("feed", ["all_at_once", "just_in_time"])
def test_fed_voice_equals_the_preloaded_voice_bit_for_bit(channels, rate, feed):
    """A source in a sub-mixer (behind a Filter) and one on the main mixer, at the mixer's rate (plain ring reads) and at other rates (the
    ResampledSource of src/source/converted.rs:15-45 pulls 512-frame chunks from the ring as it pulls them from a file source), mono and
    stereo, ragged block sizes, a volume and a panning event, a late start time. Fed in one piece up front, or piecewise ahead of every
    write. The comparison voice is preloaded with the same PCM and `source_rate` = its own rate (no embedded resampling: the same adapter chain)."""
    sizes = [1024, 333, 1, 700, 1024, 64, 1024, 1024, 511, 1024] * 2
    seconds = 0.25
    bufs = [pcm_for(4, rate, seconds, channels), pcm_for(17, rate, seconds * 0.8, channels)]

    def build(g, fed):
        m = g.add_mixer()
        g.add_effect(m, _capi.FX_FILTER, params={"cuto": 3000.0})
        vs = []
        for k, (mixer, start) in enumerate(((m, 0), (0, 1500))):
            if fed:
                vs.append(g.add_stream_voice(mixer, channels, rate, 65536, volume=0.7, panning=0.2 - 0.5 * k, start_time=start))
            else:
                vs.append(g.add_voice(mixer, bufs[k], channels, rate, volume=0.7, panning=0.2 - 0.5 * k, start_time=start, source_rate=rate, fade_out_seconds=-1.0))
        return vs

    outs = []
    for fed in (True, False):
        g = graph()
        vs = build(g, fed)
        cursor = [0, 0]
        if fed and feed == "all_at_once":
            for k in range(2):
                g.feed_voice(vs[k], bufs[k])
                g.end_stream_voice(vs[k])
        chunks, pos = [], 0
        for b, n in enumerate(sizes):
            if fed and feed == "just_in_time":
                # ahead of every write: what n output frames can consume at this rate, the resampler's 512-frame read-ahead and a margin
                want = int(np.ceil(n * rate / SR)) + 512 + 8
                for k in range(2):
                    total = bufs[k].size // channels
                    have = cursor[k] - g.stream_voice_consumed(vs[k]) if cursor[k] else 0
                    take = max(0, min(total - cursor[k], want - have))
                    if take:
                        g.feed_voice(vs[k], bufs[k][cursor[k] * channels:(cursor[k] + take) * channels])
                        cursor[k] += take
                        if cursor[k] == total:
                            g.end_stream_voice(vs[k])
            if b == 3:
                g.set_voice_volume(vs[0], 0.3, pos + 100)
                g.set_voice_panning(vs[1], -0.8, pos + 50)
            o = np.zeros(2 * n, np.float32)
            assert g.write(o, pos) in (0, 2 * n)
            chunks.append(o)
            pos += n
        outs.append(np.concatenate(chunks))
        assert g.device_errors() == 0
    assert np.array_equal(outs[0], outs[1]), int(np.count_nonzero(outs[0] != outs[1]))
    assert np.abs(outs[0]).max() > 1e-2
    if rate == SR:    # (behind a ResampledSource an ended source leaves a stale input tail that the reference keeps resampling: never silent — and equal here too)
        assert np.abs(outs[0][-2048:]).max() == 0.0


@pytest.mark.parametrize("channels,rate", [(2, 44100), (1, 32000), (2, 48000), (1, 96000)])
def test_fed_voice_against_the_oracle(channels, rate):
    """The same comparison against the CPU oracle directly: a host-fed source holding a file's PCM must sound like the oracle's file source that
    runs at its own rate (`source_rate` = the file's: no embedded resampling, ConvertedSource's ResampledSource and the channel mapper behind
    it) — through the end of the data, where a ResampledSource keeps replaying its stale input range and the mixer keeps calling a source it
    has marked inactive until its write ends. Ragged blocks, a volume event inside a block, two sources that end at
    different times."""
    import oracle

    sizes = [1024, 333, 1, 700, 1024, 64, 1024, 511, 1024, 1024, 900, 1024]
    bufs = [pcm_for(4, rate, 0.12, channels), pcm_for(17, rate, 0.07, channels)]

    def render(g, fed):
        m = g.add_mixer()
        vs = []
        for k, mixer in enumerate((m, 0)):
            if fed:
                v = g.add_stream_voice(mixer, channels, rate, 65536, volume=0.7, panning=0.2 - 0.5 * k)
                g.feed_voice(v, bufs[k])
                g.end_stream_voice(v)
            else:
                v = g.add_voice(mixer, bufs[k], channels, rate, volume=0.7, panning=0.2 - 0.5 * k, source_rate=rate, fade_out_seconds=-1.0)
            vs.append(v)
        chunks, pos = [], 0
        for b, n in enumerate(sizes):
            if b == 3:
                g.set_voice_volume(vs[0], 0.3, pos + 100)
            o = np.zeros(2 * n, np.float32)
            g.write(o, pos)
            chunks.append(o)
            pos += n
        return np.concatenate(chunks)

    a = render(graph(), True)
    b = render(oracle.OracleGraph(SR, 2, 1024), False)
    d = a.astype(np.float64) - b
    assert np.abs(b).max() > 1e-2
    assert float(np.sqrt(np.mean(d * d))) <= 1e-6 and float(np.abs(d).max()) <= 1e-5, (float(np.abs(d).max()), int(np.flatnonzero(np.abs(d) > 1e-6)[:1][0]) // 2 if np.any(np.abs(d) > 1e-6) else None)


def test_stream_voice_underrun_ring_full_stop_and_end():
    """A short ring read is a source that delivered less: the rest of the block is silent and the voice carries on with the next feed; the ring
    refuses a feed it has no room for (PG_ERR_QUEUE_FULL, nothing taken) until the device's progress has been collected; stop_voice ends the
    stream voice; once ended and played out the main mixer has nothing left and write returns 0 (src/source/mixed.rs:664-670,715)."""
    import phonic_amd

    g = graph()
    v = g.add_stream_voice(0, 2, SR, 2048)
    tone = pcm_for(3, SR, 0.2, 2)[:-2]
    out = np.zeros(2048, np.float32)
    g.feed_voice(v, tone[:2 * 300])
    assert g.write(out, 0) == 2048
    assert np.array_equal(out[:600], tone[:600]) and np.all(out[600:] == 0.0)           # 300 frames were there, the rest of the block is silent
    g.feed_voice(v, tone[2 * 300:2 * 1324])
    assert g.write(out, 1024) == 2048
    assert np.array_equal(out, tone[600:600 + 2048])                                      # ... and the voice carries on where it was
    assert g.stream_voice_consumed(v) == 1324
    g.feed_voice(v, tone[2 * 1324:2 * (1324 + 2048)])                                      # exactly the capacity
    with pytest.raises(phonic_amd.PhonicError) as ei:
        g.feed_voice(v, tone[:2])
    assert ei.value.code == _capi.PG_ERR_QUEUE_FULL
    assert g.write(out, 2048) == 2048 and np.array_equal(out, tone[2 * 1324:2 * 1324 + 2048])
    with pytest.raises(phonic_amd.PhonicError):
        g.feed_voice(v, tone[:2 * 1100])                                                    # still full: the host has not collected the progress
    assert g.stream_voice_consumed(v) == 2348
    g.feed_voice(v, tone[:2 * 1000])                                                        # room again (1024 frames were read)
    g.stop_voice(v, 3072 + 10)
    assert g.write(out, 3072) == 2048
    assert np.array_equal(out[:20], tone[2 * 2348:2 * 2348 + 20]) and np.all(out[20:] == 0.0)     # stopped 10 frames into the block
    assert not g.is_voice_playing(v)
    assert g.write(out, 4096) == 0                                                          # nothing left to play
    # a voice that is ended plays out what was fed, then the mixer is empty
    v2 = g.add_stream_voice(0, 1, SR, 4096, start_time=5120)
    g.feed_voice(v2, tone[0:1000:2].copy())
    g.end_stream_voice(v2)
    with pytest.raises(phonic_amd.PhonicError) as ei:
        g.feed_voice(v2, tone[:10])
    assert ei.value.code == _capi.PG_ERR_STATE
    assert g.write(out, 5120) == 2048
    assert np.array_equal(out[0:1000:2], tone[0:1000:2]) and np.array_equal(out[1:1000:2], tone[0:1000:2]) and np.all(out[1000:] == 0.0)
    assert g.write(out, 6144) == 0
    with pytest.raises(phonic_amd.PhonicError) as ei:
        g.feed_voice(9999, tone[:2])
    assert ei.value.code == _capi.PG_ERR_NOT_FOUND


def test_seek_and_speed_are_refused_for_host_fed_voices():
    """seek and speed exist on FilePlaybackHandle only (the reference's synth / streamed sources have neither): a host-fed voice refuses both
    with a parameter error — on the plain graph and through the sharded handle — and plays on undisturbed (the device once rewound the
    ring's read position over stale frames, and `consumed` went backwards); a mixer that is removed takes its fed voice's ring with it."""
    import phonic_amd
    from phonic_amd.graph import ShardedGraph

    tone = pcm_for(5, SR, 0.2, 2)[:-2]
    for g in (graph(), ShardedGraph([0, 0], SR, 2, 1024)):
        v = g.add_stream_voice(0, 2, SR, 4096)
        out = np.zeros(2048, np.float32)
        g.feed_voice(v, tone[:2 * 2048])
        assert g.write(out, 0) == 2048 and np.array_equal(out, tone[:2048])
        for call in (lambda: g.seek_voice(v, 0.0, 1024), lambda: g.set_voice_speed(v, 2.0, 1024), lambda: g.set_voice_speed(v, 0.5, 1024, glide=12.0)):
            with pytest.raises(phonic_amd.PhonicError) as ei:
                call()
            assert ei.value.code == _capi.PG_ERR_PARAMETER
        assert g.write(out, 1024) == 2048 and np.array_equal(out, tone[2048:4096])     # ... and carries on where it was
        assert g.stream_voice_consumed(v) == 2048
        g.set_voice_volume(v, 0.5, 2048)                                                # (what the handle of any source takes)
        g.feed_voice(v, tone[2 * 2048:2 * 3072])
        assert g.write(out, 2048) == 2048 and np.abs(out).max() > 1e-3
    g = graph()
    m = g.add_mixer()
    v = g.add_stream_voice(m, 2, SR, 2048)
    g.feed_voice(v, tone[:2 * 512])
    g.remove_mixer(m)
    with pytest.raises(phonic_amd.PhonicError) as ei:
        g.feed_voice(v, tone[:2 * 512])
    assert ei.value.code == _capi.PG_ERR_NOT_FOUND


def test_fed_voices_in_super_block_launches_equal_single_block_launches():
    """Host-fed voices are rendered by the fast kernels in steady state (the ring read is a copy), so a write of several blocks takes them through
    one super-block launch: the result must equal block-by-block launches bit for bit — voices that end inside the call, one that underruns
    (fed less than the call consumes, not ended), one that starts inside it, and a reverb unit next to them."""
    def render(max_blocks):
        g = graph()
        g.set_max_blocks_per_launch(max_blocks)
        rng = np.random.default_rng(5)
        vs = []
        for i in range(12):
            m = g.add_mixer()
            if i % 3 == 0:
                g.add_effect(m, _capi.FX_REVERB, reverb_seeds=workloads.reverb_seeds(i))
            elif i % 3 == 1:
                g.add_effect(m, _capi.FX_FILTER, params={"cuto": 2500.0})
            ch = 1 if i % 4 == 3 else 2
            v = g.add_stream_voice(m, ch, SR, 32768, volume=0.5, panning=-0.4 + 0.07 * i, start_time=0 if i != 5 else 3000)
            frames = int(rng.integers(2000, 20000))
            g.feed_voice(v, pcm_for(i, SR, 0.5, ch)[: frames * ch])
            if i != 7:
                g.end_stream_voice(v)       # voice 7 underruns and stays
            vs.append(v)
        out = np.zeros(8 * 1024 * 2, np.float32)
        chunks = []
        for k in range(3):                   # the first call settles into the steady state, the later ones are super-block launches
            assert g.write(out, k * 8 * 1024) == out.size
            chunks.append(out.copy())
        consumed = [g.stream_voice_consumed(v) for v in vs]
        g.close()
        return np.concatenate(chunks), consumed

    a, ca = render(8)
    b, cb = render(1)
    assert np.abs(a).max() > 1e-3
    assert np.array_equal(a, b)
    assert ca == cb


def test_feed_and_write_allocate_nothing():
    """The rings are reserved by add_stream_voice; feeding copies into pinned memory, the write moves it with asynchronous copies on the caller's
    stream: no allocation, no release, no host wait, no blocking copy (pg_debug_hip_calls)."""
    import torch
    from phonic_amd.graph import hip_calls

    g = graph()
    m = g.add_mixer()
    g.add_effect(m, _capi.FX_REVERB, reverb_seeds=workloads.reverb_seeds(1))
    v = g.add_stream_voice(m, 2, 44100, 1 << 16)
    v2 = g.add_stream_voice(0, 1, SR, 1 << 16)
    tone = pcm_for(6, 44100, 1.0, 2)
    stream = torch.cuda.Stream()
    bus = torch.zeros(2048, device="cuda:0")
    g.feed_voice(v, tone[:2 * 4096]); g.feed_voice(v2, tone[:4096:2].copy())
    assert g.write_device(bus.data_ptr(), 2048, 0, stream.cuda_stream) == 2048        # (the first write after a change uploads the topology)
    before = hip_calls()
    for b in range(1, 9):
        g.feed_voice(v, tone[2 * 4096 * b // 4:2 * 4096 * (b + 1) // 4] if b < 3 else tone[:2 * 900])
        g.feed_voice(v2, tone[:2000:2].copy())
        assert g.write_device(bus.data_ptr(), 2048, b * 1024, stream.cuda_stream) == 2048
    assert hip_calls() == before
    stream.synchronize()
    assert float(bus.abs().max()) > 1e-3 and g.device_errors() == 0


def test_fed_voices_on_the_sharded_handle():
    """pg_sharded_add_stream_voice / feed_voice / end_stream_voice: host-fed sources are placed and fed like any other source of the sharded mixer; three
    shards (on one device) against the plain graph with the same feeds (the f32 sum over shards reassociates: tolerance)."""
    from phonic_amd.graph import Graph, ShardedGraph

    N, blocks = 1024, 10
    bufs = [pcm_for(5 + k, 44100 if k % 2 else SR, 0.15, 2 - (k % 2)) for k in range(5)]
    outs = []
    for g in (ShardedGraph([0, 0, 0], SR, 2, N), Graph(SR, 2, N, 0)):
        vs = []
        for k, b in enumerate(bufs):
            m = g.add_mixer() if k < 3 else 0
            if k < 3:
                g.add_effect(m, _capi.FX_FILTER, params={"cuto": 2000.0 + 500 * k})
            vs.append(g.add_stream_voice(m, 2 - (k % 2), 44100 if k % 2 else SR, 1 << 15, volume=0.5))
        g.add_effect(0, _capi.FX_COMPRESSOR, params={"thrs": -30.0, "rato": 20.0, "knee": 0.0, "gain": 0.0})
        for k, b in enumerate(bufs):
            g.feed_voice(vs[k], b)
            g.end_stream_voice(vs[k])
        o = np.zeros((blocks, 2 * N), np.float32)
        for blk in range(blocks):
            assert g.write(o[blk], blk * N) == 2 * N
        outs.append(o.reshape(-1))
        assert g.device_errors() == 0
        assert g.stream_voice_consumed(vs[0]) == bufs[0].size // 2
    d = outs[0].astype(np.float64) - outs[1].astype(np.float64)
    assert float(np.sqrt(np.mean(d * d))) <= 1e-6 and float(np.abs(d).max()) <= 2e-5
    assert np.abs(outs[1]).max() > 1e-2
