"""The resident route (aacg_pipeline_*: bytes in, PCM out, several batches in flight) and the bounded host waits, through the
C ABI on a real MI355X.

Parity: the committed .aac streams were decoded by the reference itself (tests/golden/streams/*.refpcm, the output of its
readChunk(), src/decoder.js:125-216); the pipeline — device parser, kept plan refreshed on the device, aacg_decode_pipelined,
PCM down — reproduces them within 1e-5 RMS / 5e-6 of the signal from the same bytes, whatever the batching, for mono, stereo,
5.1 (SCE + CPE + CPE + LFE: the layout is learnt from the stream's first frame) and a stream with coupling elements (parsed and
dropped, as the reference executes them).  Internal consistency that must hold bit for bit: batches in flight on three lanes
against the same batches one at a time on one lane."""
import json
import os
import time

import numpy as np
import pytest

import aacgpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMS = os.path.join(ROOT, "tests", "golden", "streams")
CASES = json.load(open(os.path.join(STREAMS, "manifest.json")))


def adts_frame_table(data):
    out, off = [], 0
    while off + 7 <= len(data):
        assert data[off] == 0xFF and (data[off + 1] & 0xF0) == 0xF0
        length = ((int(data[off + 3]) & 3) << 11) | (int(data[off + 4]) << 3) | (int(data[off + 5]) >> 5)
        out.append((off, length))
        off += length
    return np.array(out, aacgpu.PARSE_FRAME_DTYPE)


def load(case):
    data = np.fromfile(os.path.join(STREAMS, case["name"] + ".aac"), np.uint8)
    table = adts_frame_table(data)
    assert len(table) == case["frames"]
    return data, table, np.fromfile(os.path.join(STREAMS, case["name"] + ".refpcm"), np.float32)


def close_to(pcm, ref):
    assert np.isfinite(pcm).all()
    d = pcm.astype(np.float64) - ref
    err, sig = float(np.sqrt(np.mean(d * d))), float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    assert sig > 1e-3 and err < 1e-5 and err <= 5e-6 * sig, (err, sig)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("split", [0, 2, 1], ids=["whole", "twos", "ones"])
def test_pipeline_decodes_the_reference_streams(case, split):
    """One stream, slot 0: the whole stream as one batch, in batches of two frames (the last one shorter: another shape, another
    plan, the first one goes stale) and frame by frame — three batches in flight each time; equal to the reference's PCM."""
    data, table, refpcm = load(case)
    C, n = case["channels"], case["frames"]
    p = aacgpu.Pipeline(channels=C, max_streams=2, max_frames=max(n, 16), sample_index=case["sampleIndex"])
    step = split or n
    tickets = []
    for a in range(0, n, step):
        tickets.append((a, min(n, a + step), p.submit(data, table[a:a + step], [0], min(n, a + step) - a)))
    pcm = np.zeros(n * 1024 * C, np.float32)
    for a, b, t in tickets:
        got, res, refused = p.collect(t)
        assert refused == 0 and not res["status"].any(), (case["name"], a, res)
        pcm[a * 1024 * C:b * 1024 * C] = got
    close_to(pcm, refpcm)
    elems, kept = p.stream_layout(0)
    assert sum(elems[:kept]) == C or case["name"] == "cce96", (elems, kept)      # cce96: SCE + CPE; its coupling element is not an output element
    p.close()


@pytest.mark.parametrize("lanes,F", [(3, 5), (6, 2)], ids=["3 lanes", "6 lanes (two priority levels)"])
def test_pipeline_batches_in_flight_equal_batches_one_at_a_time(lanes, F):
    """64 streams (the stereo stream at 64 different starting frames) x as many batches as lanes: every batch submitted before the
    first is collected, against one lane and synchronous calls — bit for bit; and the streams that start at frame 0 equal the
    reference.  Lanes beyond the fourth ride streams of another priority level (other hardware queues)."""
    case = CASES[0]
    data, table, refpcm = load(case)
    n, S, B = case["frames"], 64, lanes
    starts = [s % (n - F * B + 1) for s in range(S)]
    def batch(b):
        return np.concatenate([table[starts[s] + b * F: starts[s] + (b + 1) * F] for s in range(S)])
    a = aacgpu.Pipeline(channels=2, max_streams=S, max_frames=F, lanes=lanes)
    o = aacgpu.Pipeline(channels=2, max_streams=S, max_frames=F, lanes=1)
    pinned = [a.pinned(S * F * 2048, np.float32) for _ in range(B)]
    tickets = [a.submit(data, batch(b), np.arange(S), F, pcm=pinned[b]) for b in range(B)]
    for b in range(B):
        got, res, refused = a.collect(tickets[b])
        want, res1, refused1 = o.decode(data, batch(b), np.arange(S), F)
        assert refused == 0 and refused1 == 0
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), b
        for s in range(S):
            if starts[s] == 0:
                ref = refpcm[b * F * 2048:(b + 1) * F * 2048]          # (a batch of two frames may be a quiet stretch: per sample, not by rms)
                assert np.abs(got.reshape(S, F * 2048)[s].astype(np.float64) - ref).max() <= 1e-5 * max(1.0, 4.0 * float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))))
    a.close()
    o.close()


def test_pipeline_at_the_benchmarked_batch_size_equals_the_reference_on_every_stream():
    """256 streams x 16 frames per batch (BASELINE config 2's batch: 4096 stereo frames), the next batch (the stream's last two
    frames) in flight behind it, into page-locked memory: every one of the 256 streams equals what the reference decoded."""
    case = CASES[0]
    data, table, refpcm = load(case)
    n, S = case["frames"], 256
    p = aacgpu.Pipeline(channels=2, max_streams=S, max_frames=16)
    shapes = [(0, 16), (16, n - 16)]
    bufs = [p.pinned(S * F * 2048, np.float32) for _, F in shapes]
    tickets = [p.submit(data, np.tile(table[a:a + F], S), np.arange(S), F, pcm=bufs[i]) for i, (a, F) in enumerate(shapes)]
    for (a, F), t in zip(shapes, tickets):
        pcm, res, refused = p.collect(t)
        assert refused == 0 and not res["status"].any()
        pcm = pcm.reshape(S, F * 2048)
        assert np.array_equal(pcm.view(np.uint32), np.tile(pcm[:1].view(np.uint32), (S, 1)))      # 256 slots, one stream: one result
        close_to(pcm[0], refpcm[a * 2048:(a + F) * 2048])
    p.close()


def test_large_page_locked_blocks_take_the_pcm_like_small_ones():
    """aacg_host_alloc: blocks of 8 MiB and more are fresh huge pages faulted in by a few threads and registered (round 6: 0.6 ms
    per 32 MiB instead of 3.5-6); the PCM that comes down into one equals the PCM that comes down into pageable memory, the block
    is zero when handed out, and freeing both kinds works in any order."""
    case = CASES[0]
    data, table, refpcm = load(case)
    S, F = 640, 2                                            # 640 x 2 x 2048 floats = 10 MiB of PCM: a large block
    n = case["frames"]
    starts = [s % (n - F + 1) for s in range(S)]
    frames = np.concatenate([table[starts[s]: starts[s] + F] for s in range(S)])
    p = aacgpu.Pipeline(channels=2, max_streams=S, max_frames=F, lanes=2)
    big = p.pinned(S * F * 2048, np.float32)
    assert big.nbytes >= 8 << 20 and not big.any()
    big[::4099] = 7.0                                        # writable from the host, every page of it
    small = p.pinned(1024, np.float32)
    small[:] = 1.0
    got, res, refused = p.collect(p.submit(data, frames, np.arange(S), F, pcm=big))
    assert refused == 0 and not res["status"].any()
    for s in range(S):
        p.reset_stream(s)
    want, _, _ = p.decode(data, frames, np.arange(S), F)     # into a numpy array of the binding's: pageable memory
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    close_to(got.reshape(S, F * 2048)[0], refpcm[:F * 2048])
    lib = p.lib
    for ptr in list(p._pinned):                              # the large one first, then the small one
        lib.aacg_host_free(ptr)
    p._pinned = []
    p.close()


def test_a_slot_listed_twice_in_a_batch_is_an_error():
    data, table, _ = load(CASES[0])
    p = aacgpu.Pipeline(channels=2, max_streams=4, max_frames=1)
    with pytest.raises(aacgpu.AacgError):
        p.decode(data, table[0:2].copy(), np.array([1, 1]), 1)
    pcm, res, refused = p.decode(data, table[0:2].copy(), np.array([1, 2]), 1)      # the pipeline is usable afterwards
    assert refused == 0 and not res["status"].any()
    p.close()


def test_refusals_are_counted_once_when_a_kept_plan_has_gone_stale():
    """A pipeline keeps a plan per batch shape; when another shape's plan has advanced the streams since, the kept one is stale, the
    submission makes a new one and refreshes again — and must not count the batch's refused frames twice (found by
    tools/soak_resident.py in round 6)."""
    data, table, _ = load(CASES[0])
    data = data.copy()
    off, length = int(table[3]["byte_offset"]), int(table[3]["byte_length"])
    data[off + 7: off + length] = 0xFF                      # frame 3: a raw_data_block that ends at once (no CPE in it)
    p = aacgpu.Pipeline(channels=2, max_streams=1, max_frames=2, lanes=2)
    for a, F in ((0, 1), (1, 2)):                            # shape (1 frame), then shape (2 frames): the first shape's plan is stale now
        _, res, refused = p.decode(data, table[a:a + F].copy(), np.arange(1), F)
        assert refused == 0 and not res["status"].any()
    _, res, refused = p.decode(data, table[3:4].copy(), np.arange(1), 1)
    assert refused == 1, refused
    p.close()


def test_pipeline_refuses_a_frame_of_another_layout_as_a_whole():
    """A 5.1 stream whose third frame is replaced by a stereo frame: that frame is refused as a whole (AACG_PARSE_LAYOUT), the
    stream's state moves on through a silent frame, the frames before it are the reference's."""
    sur, stereo = CASES[1], CASES[0]
    d5, t5, ref5 = load(sur)
    d2, t2, _ = load(stereo)
    data = np.concatenate([d5, d2])
    table = t5.copy()
    table[2] = (len(d5) + int(t2[0]["byte_offset"]), int(t2[0]["byte_length"]))
    p = aacgpu.Pipeline(channels=6, max_streams=1, max_frames=8, sample_index=sur["sampleIndex"])
    pcm, res, refused = p.decode(data, table, [0], len(table))
    assert refused == 1 and res["status"][2] == aacgpu.PARSE_LAYOUT and not res["status"][[0, 1, 3, 4]].any()
    close_to(pcm[:2 * 1024 * 6], ref5[:2 * 1024 * 6])
    assert np.isfinite(pcm).all()
    p.close()


def test_a_stalled_stream_gives_a_timeout_and_a_dump_not_a_hang():
    """The engine's own stream busy for 400 ms, the wait limit at 50 ms: aacg_synchronize returns AACG_ERR_TIMEOUT within a
    fraction of a second and aacg_last_error says what was in flight; once the stream has drained the same call succeeds."""
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, 2, 2)
    eng.set_wait_limit_ms(50)
    eng.debug_stall(0, 400)
    t0 = time.perf_counter()
    with pytest.raises(aacgpu.AacgError) as ei:
        eng.synchronize()
    dt = time.perf_counter() - t0
    assert ei.value.code == aacgpu.ERR_TIMEOUT and dt < 0.3, (ei.value, dt)
    text = str(ei.value)
    assert "engine stream BUSY" in text and "launches issued" in text and "cross-launch cells" in text, text
    with pytest.raises(aacgpu.AacgError):
        eng.get_overlap(0, 0)                              # every wait is bounded, not only aacg_synchronize
    time.sleep(0.5)
    eng.set_wait_limit_ms(30000)
    eng.synchronize()
    assert eng.get_overlap(0, 0).shape == (1024,)
    eng.close()


def test_a_stalled_pipeline_stream_times_out_in_the_pipelined_route():
    """One of the pipeline's streams busy: aacg_synchronize (which joins the pipelined launches) times out with that stream
    named BUSY in the dump; afterwards the engine decodes as before."""
    import aacgpu_workload
    import orc
    wl = aacgpu_workload.make_batch(n_streams=2, n_frames=4, seed=5)
    eng = aacgpu.Engine(aacgpu.INPUT_QUANT_I16, 2, 2)
    eng.set_wait_limit_ms(40)
    eng.debug_stall(2, 300)
    with pytest.raises(aacgpu.AacgError) as ei:
        eng.synchronize()
    assert ei.value.code == aacgpu.ERR_TIMEOUT and "pipe stream 1 BUSY" in str(ei.value), str(ei.value)
    time.sleep(0.4)
    eng.set_wait_limit_ms(30000)
    eng.synchronize()
    pcm = eng.decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"])
    ref = orc.load().decode_batch(wl["units"], wl["q"], wl["meta"], wl["n_pcm"], np.zeros((2, 2, 1024), np.float32))
    close_to(pcm, ref)
    streams, concurrent = eng.pipeline_info()
    assert concurrent in (0, 1)
    eng.close()
