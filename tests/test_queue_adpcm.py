"""Streaming sources and compressed buffers.

  * Buffer queues -- VoiceBufferItem::mNext, LoadBufferQueue (core/voice.cpp:563-594), the buffer advance in
    Voice::mix (voice.cpp:1182-1194): voices that are not VoiceFlag::IsStatic crawl a queue of buffers of very
    different lengths (2 .. 5000 samples), loop back to the queue's head or run off its end and stop.
  * IMA4 / MS ADPCM -- LoadSamples<IMA4Data>, LoadSamples<MSADPCMData> (voice.cpp:288-484): mono and stereo
    data, block sizes from the smallest legal to > 1000 samples, as static buffers and inside a queue.  The
    product decodes a buffer once at registration (csrc/adpcm_kernels.hip); the reference decodes in every mix.

CPU: the compiled reference decodes the test encoders' blocks to exactly the samples the encoders predict
(tests/adpcm_codec.py) -- that pins the expectation.  GPU: every voice kernel against the compiled reference;
decoded samples bit-exact (unity gain, point resampler), mixes within the usual tolerance, positions, play
states, current buffers and completed-buffer counts exact after every update."""
import os

import numpy as np
import pytest

import adpcm_codec as ac
import oracle_lib as ol

ADPCM_CASES = [(0, 65, 1), (1, 64, 1), (0, 9, 2), (1, 500, 2), (0, 1017, 1), (1, 4, 1), (1, 2046, 2)]


def _ref():
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference (oracle/_ref)")
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    return L


def decoded_through(lib, typ, spb, ch, data, n, **kw):
    """the buffer's samples as a voice hears them: unity gain, step 1.0, point resampler -> dry line c"""
    sc = lib.make_scene(num_dry=5, num_real=0, num_sends=0, num_slots=0, wet_channels=4, hrtf=False, **kw)
    buf = sc.add_buffer_adpcm(data, typ, ch, spb, n)
    assert buf >= 0
    if ch == 1:
        v = sc.add_voice(buf, looping=False, position=0)
        sc.set_params(v, ol.make_voice_params(65536, ol.RS_POINT, dry_gains=[1, 0, 0, 0, 0]))
    else:
        v = sc.add_ambi_voice(buf, 2, looping=False, position=0)
        for c in range(2):
            sc.set_channel_params(v, c, ol.make_voice_params(65536, ol.RS_POINT, dry_gains=[c == 0, c == 1, 0, 0, 0]))
    got = []
    for _ in range((n + 999) // 1000):
        sc.mix(1000)
        got.append(sc.dry()[:ch, :1000].T.copy())
    sc.close()
    return np.concatenate(got)[:n]


@pytest.mark.parametrize("typ,spb,ch", ADPCM_CASES)
def test_reference_decodes_what_the_encoders_predict(typ, spb, ch):
    L = _ref()
    n = 4100
    data, dec = (ac.encode_ima4 if typ == 0 else ac.encode_msadpcm)(ac.test_signal(n, ch, 5 + spb), spb)
    want = dec.astype(np.float32).reshape(n, ch) / np.float32(32768.0)
    assert np.abs(want).max() > 0.2
    assert np.array_equal(decoded_through(L, typ, spb, ch, data, n), want)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("typ,spb,ch", ADPCM_CASES)
def test_gpu_decode_is_bit_exact(typ, spb, ch, mode):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    n = 4100
    data, dec = (ac.encode_ima4 if typ == 0 else ac.encode_msadpcm)(ac.test_signal(n, ch, 5 + spb), spb)
    want = dec.astype(np.float32).reshape(n, ch) / np.float32(32768.0)
    api = oalgpu.Api(oalgpu.MATH_EXACT if mode == "exact" else oalgpu.MATH_FAST)
    assert np.array_equal(decoded_through(api, typ, spb, ch, data, n, max_voices=4), want)


# ---- scenes: queues of PCM and ADPCM buffers, static ADPCM voices ------------------------------------------
QUEUE_A = (300, 1500, 64, 2, 5000)            # buffer lengths, linked in this order
QUEUE_B = (777, 40, 3000)
TODO = (1024, 1024, 600, 1024, 1024, 1024)


def run(lib, mhr, hrtf, sends, nvoices=16):
    if hrtf:
        lib.hrtf_load(mhr)
    kw = dict(max_voices=nvoices + 2, max_buffers=64) if hasattr(lib, "device") else {}
    sc = lib.make_scene(num_dry=4 if hrtf else 5, num_real=2 if hrtf else 0, num_sends=sends, num_slots=2 if sends else 0,
                        wet_channels=4, hrtf=hrtf, **kw)
    rng = np.random.default_rng(77)
    if hrtf:
        cc = np.zeros((4, 128, 2), np.float32)
        cc[:, :64] = rng.uniform(-0.2, 0.2, (4, 64, 2))
        sc.set_direct_hrtf(cc, [1.0, 0.8, 0.8, 0.8], 400.0 / 48000.0, 64)
    qa = [sc.add_buffer(rng.uniform(-1, 1, n).astype(np.float32), ol.FMT_FLOAT) for n in QUEUE_A]
    qb = [sc.add_buffer(rng.integers(-30000, 30000, n).astype(np.int16), ol.FMT_SHORT) for n in QUEUE_B]
    # a queue of compressed buffers: IMA4, MS ADPCM, IMA4
    qc = []
    for i, (typ, spb, n) in enumerate([(0, 65, 1300), (1, 64, 900), (0, 9, 2000)]):
        data, _ = (ac.encode_ima4 if typ == 0 else ac.encode_msadpcm)(ac.test_signal(n, 1, 40 + i), spb)
        qc.append(sc.add_buffer_adpcm(data, typ, 1, spb, n))
    for q in (qa, qb, qc):
        for a, b in zip(q, q[1:]):
            sc.link_buffers(a, b)
    # static compressed buffers with loop points in the middle of blocks
    data, _ = ac.encode_ima4(ac.test_signal(6000, 1, 50), 65)
    static_ima = sc.add_buffer_adpcm(data, 0, 1, 65, 6000, 100, 5900)
    data, _ = ac.encode_msadpcm(ac.test_signal(6000, 1, 51), 64)
    static_ms = sc.add_buffer_adpcm(data, 1, 1, 64, 6000, 333, 4444)
    starts = {}
    for v in range(nvoices):
        kind = v % 8
        pos, frac = (v * 131) % 250, (v * 977) % 65536
        if kind in (0, 1):
            sc.add_queue_voice(qa[0], looping=kind == 0, position=pos, frac=frac)
        elif kind in (2, 3):
            sc.add_queue_voice(qb[0], looping=kind == 2, position=pos, frac=frac)
        elif kind in (4, 5):
            sc.add_queue_voice(qc[0], looping=kind == 4, position=pos, frac=frac)
        elif kind == 6:
            sc.add_voice(static_ima, looping=True, position=pos * 20, frac=frac)
        else:
            sc.add_voice(static_ms, looping=v % 16 == 7, position=pos * 20, frac=frac)
        r = np.random.default_rng(300 + v)
        step = [65536, 100000, 30000, 230000, 70001][v % 5]         # up to 3.5 source samples per output sample
        snd = [(i % 2, r.uniform(0.05, 0.3, 4), ol.default_filter(active=(v + i) % 2, gain_hf=0.6)) for i in range(sends)]
        if hrtf:
            p = ol.make_voice_params(step, ol.RS_BSINC24, hrtf=(np.arcsin(r.uniform(-1, 1)), r.uniform(-np.pi, np.pi), 2.0, 0.0, 0.1),
                                     direct_filter=ol.default_filter(active=v % 2, gain_hf=0.5), sends=snd)
        else:
            p = ol.make_voice_params(step, ol.RS_BSINC24, dry_gains=r.uniform(0, 0.2, 5),
                                     direct_filter=ol.default_filter(active=v % 2, gain_hf=0.5), sends=snd)
        sc.set_params(v, p)
    sc.set_start_delay(9, 500)                                  # a streaming voice with a delayed start
    out, ints = [], []
    for k, n in enumerate(TODO):
        if k == 3:
            sc.set_state(0, ol.VOICE_STOPPING)                  # a streaming voice stopped mid-queue
        sc.mix(n, post_process=hrtf)
        parts = [sc.dry()[:, :n].ravel()]
        if hrtf:
            parts.append(sc.hrtf_accum().ravel())
        for s in range(2 if sends else 0):
            parts.append(sc.wet(s)[:, :n].ravel())
        out.append(np.concatenate(parts).astype(np.float64))
        row = []
        for v in range(nvoices):
            s = sc.voice_state(v)
            cur, done = sc.queue_state(v)
            row.append((s.play_state, s.position, s.position_frac, s.has_buffer, cur if s.has_buffer else -1, done))
        ints.append(row)
    sc.close()
    return out, ints


CASES = {
    "hrtf fast (wavefront kernel, matrix-pipe FIR)": dict(hrtf=True, sends=0, exact=False, flags=0),
    "hrtf fast (wavefront kernel, packed-VALU FIR)": dict(hrtf=True, sends=0, exact=False, flags=1),
    "hrtf fast + sends (stream rows)": dict(hrtf=True, sends=2, exact=False, flags=0),
    "hrtf exact (generic kernel)": dict(hrtf=True, sends=0, exact=True, flags=0),
    "dry lines fast (stream rows)": dict(hrtf=False, sends=0, exact=False, flags=0),
    "dry lines + sends exact": dict(hrtf=False, sends=2, exact=True, flags=0),
    "dry lines + sends fast (rows in LDS)": dict(hrtf=False, sends=2, exact=False, flags=0),
    "dry lines + sends fast (a wavefront per slice)": dict(hrtf=False, sends=2, exact=False, flags=128),      # OALGPU_CTX_SLICE_LINES
    "dry lines + sends fast (stream rows)": dict(hrtf=False, sends=2, exact=False, flags=8),                  # OALGPU_CTX_STREAM_ROWS
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_streaming_and_compressed_voices_match_the_reference(case, synth_mhr):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    L = _ref()
    cfg = CASES[case]
    want, wi = run(L, synth_mhr, cfg["hrtf"], cfg["sends"])
    got, gi = run(oalgpu.Api(oalgpu.MATH_EXACT if cfg["exact"] else oalgpu.MATH_FAST, ctx_flags=cfg["flags"]), synth_mhr, cfg["hrtf"], cfg["sends"])
    for k in range(len(TODO)):
        assert gi[k] == wi[k], (case, k, [(v, a, b) for v, (a, b) in enumerate(zip(gi[k], wi[k])) if a != b][:4])
        err = np.abs(got[k] - want[k]).max()
        assert err <= 2e-5 * np.abs(want[k]).max() + 1e-7, (case, k, err)
    # the scene exercises what it claims: buffers completed, a queue ran out, a looping queue wrapped
    assert max(r[5] for r in wi[-1]) >= 5 and any(r[0] == ol.VOICE_STOPPED for r in wi[-1])
    assert wi[-1][8][0] == ol.VOICE_PLAYING and wi[-1][8][5] >= len(QUEUE_A)


def test_reference_queue_scene_is_meaningful(synth_mhr):
    """CPU: the streaming scene on the compiled reference alone -- buffers complete, queues end and wrap"""
    L = _ref()
    _, wi = run(L, synth_mhr, False, 0)
    assert max(r[5] for r in wi[-1]) >= 5 and any(r[0] == ol.VOICE_STOPPED for r in wi[-1])
    assert wi[-1][8][0] == ol.VOICE_PLAYING and wi[-1][8][5] >= len(QUEUE_A)
    assert wi[0][9][1] != wi[1][9][1]
