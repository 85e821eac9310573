"""Delayed start (Voice::mStartTime, core/voice.cpp:1023-1046): a voice scheduled to start `d` output samples
ahead mixes nothing while d >= samplesToDo, then samplesToDo - outPos samples at output position
outPos = d mod ..., and from then on plays normally; stopped before it started it just becomes Stopped.
Every voice kernel (EXACT / FAST generic, wavefront per voice with and without stream rows, workgroup per
voice) against the reference: buses within the usual tolerance (EXACT single-voice: bit-exact), positions
and play states exact after every update."""
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
DELAYS = {1: 1, 2: 37, 3: 700, 5: 1023, 6: 1024, 7: 1500, 9: 2500, 10: 64, 12: 5}     # voice -> samples ahead
TODO = (1024, 700, 1024, 1024, 300)


def run(lib, mhr, hrtf, sends, nvoices=14, stop_unstarted=(7,)):
    if hrtf:
        lib.hrtf_load(mhr)
    kw = dict(max_voices=nvoices) if hasattr(lib, "device") else {}
    sc = lib.make_scene(num_dry=4 if hrtf else 5, num_real=2 if hrtf else 0, num_sends=sends, num_slots=2 if sends else 0,
                        wet_channels=4, hrtf=hrtf, **kw)
    rng = np.random.default_rng(21)
    if hrtf:
        cc = np.zeros((4, 128, 2), np.float32)
        cc[:, :64] = rng.uniform(-0.2, 0.2, (4, 64, 2))
        sc.set_direct_hrtf(cc, [1.0, 0.8, 0.8, 0.8], 400.0 / 48000.0, 64)
    buf = sc.add_buffer(rng.uniform(-1, 1, 9000).astype(np.float32), ol.FMT_FLOAT, loop_start=100, loop_end=8900)
    for v in range(nvoices):
        sc.add_voice(buf, looping=v % 3 != 2, position=(v * 977) % 8000, frac=(v * 131) % 65536)
        r = np.random.default_rng(100 + v)
        snd = [(i % 2, r.uniform(0.05, 0.3, 4), ol.default_filter(active=(v + i) % 2, gain_hf=0.6)) for i in range(sends)]
        if hrtf:
            p = ol.make_voice_params([60211, 70000, 48000][v % 3], ol.RS_BSINC24,
                                     hrtf=(np.arcsin(r.uniform(-1, 1)), r.uniform(-np.pi, np.pi), 2.0, 0.0, 0.1),
                                     direct_filter=ol.default_filter(active=v % 2, gain_hf=0.5), sends=snd)
        else:
            p = ol.make_voice_params([60211, 70000, 48000][v % 3], ol.RS_BSINC24, dry_gains=r.uniform(0, 0.2, 5),
                                     direct_filter=ol.default_filter(active=v % 2, gain_hf=0.5), sends=snd)
        sc.set_params(v, p)
        if v in DELAYS:
            sc.set_start_delay(v, DELAYS[v])
    out, ints = [], []
    for k, n in enumerate(TODO):
        if k == 1:
            for v in stop_unstarted:                        # still waiting for its start: becomes Stopped without a sound
                sc.set_state(v, ol.VOICE_STOPPING)
        sc.mix(n, post_process=hrtf)
        parts = [sc.dry()[:, :n].ravel()]
        if hrtf:
            parts.append(sc.hrtf_accum().ravel())
        for s in range(2 if sends else 0):
            parts.append(sc.wet(s)[:, :n].ravel())
        out.append(np.concatenate(parts).astype(np.float64))
        st = [sc.voice_state(v) for v in range(nvoices)]
        ints.append([(s.play_state, s.position, s.position_frac, s.has_buffer) for s in st])
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


@pytest.mark.parametrize("case", list(CASES))
def test_delayed_voices_match_the_reference(case, synth_mhr):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference")
    cfg = CASES[case]
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    want, wi = run(L, synth_mhr, cfg["hrtf"], cfg["sends"])
    got, gi = run(oalgpu.Api(oalgpu.MATH_EXACT if cfg["exact"] else oalgpu.MATH_FAST, ctx_flags=cfg["flags"]), synth_mhr, cfg["hrtf"], cfg["sends"])
    for k in range(len(TODO)):
        assert gi[k] == wi[k], (case, k, [(v, a, b) for v, (a, b) in enumerate(zip(gi[k], wi[k])) if a != b][:4])
        err = np.abs(got[k] - want[k]).max()
        assert err <= 2e-5 * np.abs(want[k]).max() + 1e-7, (case, k, err)
    # the delays did something: voice 9 (2500 samples ahead) is silent for two updates and sounds in the third
    assert wi[0][9][1] == wi[1][9][1] and wi[2][9][1] != wi[1][9][1]
    assert wi[-1][7][0] == ol.VOICE_STOPPED


def test_machine_filling_hrtf_scene_with_filtered_sends_matches_the_reference(synth_mhr):
    """4096 voices (the 16-wavefront form of csrc/voice_wave16.hip with sends: every wavefront's send rows -- half of them through the
    send's own filter pair -- go out as stream rows and StreamRowsMixKernel mixes them) against the reference: the same scene as
    above at the size where the kernel's register budget is 112, delays, stops and odd update lengths included"""
    import oalgpu
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference")
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    n = 4096
    want, wi = run(L, synth_mhr, True, 2, nvoices=n)
    api = oalgpu.Api(oalgpu.MATH_FAST)
    got, gi = run(api, synth_mhr, True, 2, nvoices=n)
    for k in range(len(TODO)):
        assert gi[k] == wi[k], (k, [(v, a, b) for v, (a, b) in enumerate(zip(gi[k], wi[k])) if a != b][:4])
        err = np.abs(got[k] - want[k]).max()
        assert err <= 4e-5 * np.abs(want[k]).max() + 1e-7, (k, err, np.abs(want[k]).max())
