"""Callback sources ("queue/callback sources"; AL_SOFT_callback_buffer).

The reference calls the buffer's user function in the middle of Voice::mix for exactly the blocks the voice's
position needs (core/voice.cpp:726-752), loads through LoadBufferCallback (:546-561) and drops consumed blocks
afterwards (:1155-1180).  The product calls the user function on the host before the update's voice kernel with the
same byte counts (a host mirror of the voice's integer state), and hands the storage to the voice kernels as a static,
non-looping buffer of mNumCallbackBlocks samples read from mCallbackBlockOffset.

Scenes mix callback voices beside static ones: streams that outlast the run and streams that end in the middle of an
update (the callback returns short: CallbackStopped, the voice holds the last sample, ends, fades), every PCM format,
pitches from 0.46 to 3.5 source samples per output sample, ragged update sizes, a voice stopped by the application.
Compared after every update: the mix (bit for bit with one voice in EXACT mode, the usual tolerance otherwise), play
state, fractional position, mNumCallbackBlocks, mCallbackBlockOffset, CallbackStopped -- and the number of times the
user function was called."""
import os

import numpy as np
import pytest

import oracle_lib as ol

TODO = [1024, 1024, 700, 1024, 1, 1024, 333, 1024, 1024, 1024]
STEPS = [65536, 100000, 30000, 230000, 70001, 60211]


def _ref():
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference (oracle/_ref)")
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    return L


def stream_for(v, fmt, frames):
    r = np.random.default_rng(700 + v)
    if fmt == ol.FMT_FLOAT:
        return r.uniform(-1, 1, frames).astype(np.float32)
    if fmt == ol.FMT_SHORT:
        return r.integers(-32768, 32767, frames).astype(np.int16)
    if fmt == ol.FMT_DOUBLE:
        return r.uniform(-1, 1, frames)
    if fmt == ol.FMT_INT:
        return r.integers(-2 ** 31, 2 ** 31 - 1, frames).astype(np.int32)
    return r.integers(0, 255, frames).astype(np.uint8)


FMTS = [ol.FMT_FLOAT, ol.FMT_SHORT, ol.FMT_UBYTE, ol.FMT_DOUBLE, ol.FMT_INT, ol.FMT_MULAW, ol.FMT_ALAW]


def run(lib, mhr, hrtf, nvoices, single=False, **kw):
    if hrtf:
        lib.hrtf_load(mhr)
    sc = lib.make_scene(num_dry=4 if hrtf else 5, num_real=2 if hrtf else 0, num_sends=0, num_slots=0, wet_channels=4, hrtf=hrtf, **kw)
    if hrtf:
        cc = np.zeros((4, 128, 2), np.float32); cc[:, :64] = np.random.default_rng(3).uniform(-0.2, 0.2, (4, 64, 2))
        sc.set_direct_hrtf(cc, [1.0, 0.8, 0.8, 0.8], 400.0 / 48000.0, 64)
    static = sc.add_buffer(np.random.default_rng(1).uniform(-1, 1, 6000).astype(np.float32), ol.FMT_FLOAT)
    cb_voices = []
    for v in range(nvoices):
        step = STEPS[v % len(STEPS)]
        if v % 4 == 3 and not single:
            sc.add_voice(static, looping=True, position=v * 100)          # static voices between them
        else:
            fmt = FMTS[v % len(FMTS)]
            # every third stream ends somewhere inside the run (one of them in the very first update)
            total = sum(TODO) * step // 65536
            frames = total + 5000 if v % 3 else (300 if v == 0 and not single else total // 2 + 37 * v)
            sc.add_callback_voice(stream_for(v, fmt, frames), fmt, frac=(v * 977) % 65536)
            cb_voices.append(v)
        r = np.random.default_rng(300 + v)
        if hrtf:
            p = ol.make_voice_params(step, ol.RS_BSINC24, hrtf=(np.arcsin(r.uniform(-1, 1)), r.uniform(-np.pi, np.pi), 2.0, 0.0, 0.1),
                                     direct_filter=ol.default_filter(active=v % 2, gain_hf=0.5))
        else:
            p = ol.make_voice_params(step, ol.RS_SPLINE if v % 2 else ol.RS_BSINC24, dry_gains=r.uniform(0, 0.2, 5),
                                     direct_filter=ol.default_filter(active=v % 2, gain_hf=0.5))
        sc.set_params(v, p)
    out, ints = [], []
    for k, n in enumerate(TODO):
        if k == 5 and not single and nvoices > 1:
            sc.set_state(cb_voices[1], ol.VOICE_STOPPING)            # the application stops a callback voice
        if k == 2:                                                   # a pitch change in mid-stream
            v = cb_voices[-1]
            r = np.random.default_rng(900)
            step = 41000
            if hrtf:
                p = ol.make_voice_params(step, ol.RS_BSINC24, hrtf=(0.3, -1.0, 2.0, 0.0, 0.1), direct_filter=ol.default_filter(active=0))
            else:
                p = ol.make_voice_params(step, ol.RS_BSINC24, dry_gains=r.uniform(0, 0.2, 5), direct_filter=ol.default_filter(active=0))
            sc.set_params(v, p)
        sc.mix(n, post_process=hrtf)
        parts = [sc.dry()[:, :n].ravel()]
        if hrtf:
            parts.append(sc.hrtf_accum().ravel())
        out.append(np.concatenate(parts).astype(np.float64))
        row = []
        for v in range(nvoices):
            s = sc.voice_state(v)
            cb = sc.callback_state(v) if v in cb_voices else None
            # a callback voice's integer position is relative to its storage on the device: compare the fraction,
            # the play state and whether it still has its buffer; the blocks / offset / stopped / calls come beside
            row.append((s.play_state, s.position_frac, s.has_buffer, cb) if cb is not None
                       else (s.play_state, s.position, s.position_frac, s.has_buffer))
        ints.append(row)
    sc.close()
    return out, ints


def test_reference_callback_scene_is_meaningful(synth_mhr):
    """No GPU: streams end inside the run, the user function stops being called, voices end and fade"""
    L = _ref()
    out, ints = run(L, synth_mhr, False, 12)
    assert max(np.abs(o).max() for o in out) > 0.05
    ended = [v for v in range(12) if v % 4 != 3 and ints[-1][v][0] == ol.VOICE_STOPPED]
    alive = [v for v in range(12) if v % 4 != 3 and ints[-1][v][0] == ol.VOICE_PLAYING]
    assert len(ended) >= 3 and len(alive) >= 3, (ended, alive)
    assert ints[0][0][3][2] == 1, "the 300-frame stream ran dry in the first update: CallbackStopped"
    calls = [ints[-1][v][3][3] for v in alive]
    assert min(calls) >= len(TODO) - 1, "one request per update while the stream lasts"


CASES = {
    "dry lines exact (generic kernel)": dict(hrtf=False, exact=True, flags=0),
    "dry lines fast (stream rows)": dict(hrtf=False, exact=False, flags=0),
    "hrtf fast (wavefront kernel, matrix-pipe FIR)": dict(hrtf=True, exact=False, flags=0),
    "hrtf fast (wavefront kernel, packed-VALU FIR)": dict(hrtf=True, exact=False, flags=1),
    "hrtf exact (generic kernel)": dict(hrtf=True, exact=True, flags=0),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_callback_voices_match_the_reference(case, synth_mhr):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    L = _ref()
    cfg = CASES[case]
    want, wi = run(L, synth_mhr, cfg["hrtf"], 12)
    got, gi = run(oalgpu.Api(oalgpu.MATH_EXACT if cfg["exact"] else oalgpu.MATH_FAST, ctx_flags=cfg["flags"]), synth_mhr, cfg["hrtf"], 12, max_voices=16)
    for k, (a, b) in enumerate(zip(got, want)):
        assert gi[k] == wi[k], (case, k, [(x, y) for x, y in zip(gi[k], wi[k]) if x != y][:3])
        assert np.abs(a - b).max() <= 2e-5 * max(np.abs(b).max(), 1e-3) + 1e-7, (case, k, float(np.abs(a - b).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", range(len(FMTS)))
def test_one_callback_voice_bit_exact(fmt, synth_mhr):
    """EXACT mode, one voice, a stream that ends in mid-run: every sample as the reference mixes it"""
    import oalgpu
    L = _ref()

    def one(lib, **kw):
        sc = lib.make_scene(num_dry=5, num_real=0, num_sends=0, num_slots=0, wet_channels=4, hrtf=False, **kw)
        v = sc.add_callback_voice(stream_for(40 + fmt, FMTS[fmt], 4321), FMTS[fmt], frac=12345)
        sc.set_params(v, ol.make_voice_params(60211, ol.RS_BSINC24, dry_gains=[0.5, 0.25, 0.1, 0.0, 0.3]))
        res = []
        for n in TODO[:7]:
            sc.mix(n)
            s = sc.voice_state(v)
            res.append((sc.dry()[:, :n].copy(), (s.play_state, s.position_frac, s.has_buffer, sc.callback_state(v))))
        sc.close()
        return res

    want = one(L)
    got = one(oalgpu.Api(oalgpu.MATH_EXACT), max_voices=4)
    assert want[-1][1][0] == ol.VOICE_STOPPED
    for k, ((a, ia), (b, ib)) in enumerate(zip(got, want)):
        assert ia == ib, (k, ia, ib)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (k, float(np.abs(a - b).max()))


@pytest.mark.gpu
def test_callback_sources_are_recycled(synth_mhr):
    """A callback source that has ended starts over on the same voice slot, and a voice slot that becomes another
    kind of source hands its callback entry on: buffer-table slot, device and pinned memory are reused -- a context
    with room for three buffers plays fifteen short streams, each bit for bit what the first one was."""
    import oalgpu
    api = oalgpu.Api(oalgpu.MATH_EXACT)
    sc = api.make_scene(num_dry=5, num_real=0, num_sends=0, num_slots=0, wet_channels=4, hrtf=False, max_voices=4, max_buffers=3)
    static = sc.add_buffer(np.zeros(64, np.float32), ol.FMT_FLOAT)
    stream = stream_for(3, ol.FMT_SHORT, 1500)
    first = None
    for rep in range(15):
        v = sc.add_callback_voice(stream, ol.FMT_SHORT, frac=0, voice=0 if rep else None)
        assert v == 0
        sc.set_params(0, ol.make_voice_params(65536, ol.RS_LINEAR, dry_gains=[0.5, 0.25, 0.1, 0.0, 0.3]))
        out = []
        for _ in range(3):                              # 1500 frames at unit pitch: the stream ends inside the second update
            sc.mix(1024)
            out.append(sc.dry().copy())
        assert sc.voice_state(0).play_state == ol.VOICE_STOPPED
        if first is None:
            first = out
            assert np.abs(out[0]).max() > 1e-3
        else:
            for a, b in zip(out, first):
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), rep
        if rep % 5 == 4:                                # the slot becomes a static source in between: the entry is handed on
            sc.add_voice(static, looping=False) if sc.nvoices < 2 else None
            import ctypes as C
            desc = oalgpu.VoiceDesc(static, 0, 0, 0, 44100)
            oalgpu.check(oalgpu.lib.oalgpu_voice_init(sc.h, 0, C.byref(desc)), "oalgpu_voice_init")
            sc.mix(64)
    sc.close()
