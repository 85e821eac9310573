"""The pipelined host boundary (oalgpu_voice_move_async / oalgpu_read_output_async / oalgpu_output_wait): six updates of
a config-3 scene whose moving voices are moved through it, outputs collected two updates late, against the same scene
driven through oalgpu_voice_set_params + oalgpu_read_dry with a synchronisation after every update.  A moved voice's
record carries the same direction and gain as its full parameter record, so the two must agree to the bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _moves(oalgpu, script, voices, update):
    m = np.zeros(len(voices), oalgpu.MOVE_DTYPE)
    for i, v in enumerate(voices):
        ev, az, gain = script.direction(v, update)
        m[i] = (v, ev, az, 2.0, 0.0, gain)
    return m


def test_async_boundary_matches_the_synchronous_one(synth_mhr):
    import oalgpu
    from oalgpu import synth
    import bench
    mhr = open(synth_mhr, "rb").read()
    V, updates = 512, 6
    sizes = [1024, 600, 1024, 257, 1024, 1024]      # (short updates: the lines' frames from n on come back as the bus holds them)
    outs = {}
    for mode in ("sync", "async"):
        api = oalgpu.Api(oalgpu.MATH_FAST)
        api._mhr = mhr
        sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
        allv = list(range(V))
        moving = [v for v in allv if script.is_moving(v)]
        sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
        got = []
        if mode == "sync":
            for k in range(updates):
                if k:
                    sc.set_params_batch(moving, bench.param_array(oalgpu, script, moving, k))
                sc.mix(sizes[k], post_process=True)
                got.append(sc.dry()[4:6].copy())
        else:
            tickets = []
            for k in range(updates):
                if k:
                    sc.move_async(_moves(oalgpu, script, moving, k))
                sc.mix(sizes[k], post_process=True)
                tickets.append(sc.read_output_async())
                if k >= 2:
                    got.append(sc.output_wait(tickets[k - 2]).copy())
            for t in tickets[-2:]:
                got.append(sc.output_wait(t).copy())
            with pytest.raises(oalgpu.OalgpuError):
                sc.output_wait(tickets[0])        # that slot has been reused: four tickets may be outstanding
        outs[mode] = got
        sc.close()
    assert len(outs["sync"]) == len(outs["async"]) == updates
    sounded = 0.0
    for k in range(updates):
        a, b = outs["sync"][k], outs["async"][k]
        assert a.shape == b.shape == (2, 1024)
        sounded = max(sounded, float(np.abs(a).max()))
        assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))
    assert sounded > 1e-3


def test_async_boundary_against_the_reference(synth_mhr):
    """The pipelined boundary against the ORACLE itself (not only against the synchronous boundary): ten updates of a
    config-3 scene, the moving voices handed over as 24-byte oalgpu_voice_move records (their HRIR blend indices
    evaluated on the host, installed by ApplyMovesKernel in front of the next voice kernel), the stereo lines collected
    two updates late -- compared with the compiled reference's Voice::mix + MixDirectHrtf of the same scene update by
    update, and every voice's integer state at the end (core/voice.cpp:1126-1154)."""
    import oalgpu
    import oracle_lib as ol
    from oalgpu import synth
    import bench
    from test_gpu_baseline_configs import build_reference_scene, close_to
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference (oracle/_ref)")
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    mhr = open(synth_mhr, "rb").read()
    V, updates = 512, 10
    api = oalgpu.Api(oalgpu.MATH_FAST)
    api._mhr = mhr
    sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
    osc, oscript, _ = build_reference_scene(L, synth, 3, V, synth_mhr)
    allv = list(range(V))
    moving = [v for v in allv if script.is_moving(v)]
    sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
    for v in allv:
        osc.set_params(v, oscript.fill(ol.VoiceParams(), v, 0))
    tickets, got, want = [], [], []
    for k in range(updates):
        if k:
            sc.move_async(_moves(oalgpu, script, moving, k))
            for v in moving:
                osc.set_params(v, oscript.fill(ol.VoiceParams(), v, k))
        sc.mix(1024, post_process=True)
        tickets.append(sc.read_output_async())
        if k >= 2:
            got.append(sc.output_wait(tickets[k - 2]).copy())
        osc.mix(1024, post_process=True)
        want.append(osc.dry()[4:6].copy())
    for t in tickets[-2:]:
        got.append(sc.output_wait(t).copy())
    sounded = 0.0
    for k in range(updates):
        sounded = max(sounded, close_to(got[k], want[k], f"async boundary, update {k}: stereo lines", (V, 64)))
    assert sounded > 1e-3
    for v in allv:
        g, o = sc.voice_state(v), osc.voice_state(v)
        assert (g.play_state, g.position, g.position_frac, g.has_buffer, g.fading) == \
            (o.play_state, o.position, o.position_frac, o.has_buffer, o.fading), v
    sc.close()
    osc.close()


def test_negative_hrtf_distances_are_reserved(synth_mhr):
    """include/oalgpu.h: hrtf_dist = OALGPU_HRTF_KEEP_TARGET (-1) keeps the voice's target; any other negative distance is
    rejected (the reference passes a vector norm, alc/alu.cpp:1761, :1214)"""
    import oalgpu
    from oalgpu import synth
    import bench
    mhr = open(synth_mhr, "rb").read()
    api = oalgpu.Api(oalgpu.MATH_FAST)
    api._mhr = mhr
    sc, script = bench.build_scene(oalgpu, synth, api, 3, 8, 0, mhr, 0)
    arr = bench.param_array(oalgpu, script, [0, 1], 0)
    sc.set_params_batch([0, 1], arr)
    arr[1].hrtf_dist = -1.0
    sc.set_params_batch([0, 1], arr)
    arr[1].hrtf_dist = -2.0
    with pytest.raises(oalgpu.OalgpuError):
        sc.set_params_batch([0, 1], arr)
    sc.close()


def test_voice_events_report_what_a_full_readback_would_show(synth_mhr):
    """oalgpu_voice_events_async / _wait: voices that run out of buffer (Playing -> Stopping -> Stopped) show up in the reports, with
    the state a full read-back (oalgpu_voices_readback) has at that update; voices that just play on are not reported again."""
    import oalgpu
    api = oalgpu.Api(oalgpu.MATH_FAST)
    api._mhr = open(synth_mhr, "rb").read()
    V = 300
    sc = api.make_scene(num_dry=4, num_real=2, hrtf=True, max_voices=V, max_buffers=4)
    rng = np.random.default_rng(5)
    long_buf = sc.add_buffer(rng.uniform(-1, 1, 48000).astype(np.float32), oalgpu.FMT_FLOAT, loop_start=0, loop_end=48000)
    short_buf = sc.add_buffer(rng.uniform(-1, 1, 2500).astype(np.float32), oalgpu.FMT_FLOAT)
    ending = set(range(0, V, 7))                       # these run out during the third update
    import oracle_lib as ol
    for v in range(V):
        sc.add_voice(short_buf if v in ending else long_buf, v not in ending, position=(v * 131) % 400)
        sc.set_params(v, ol.make_voice_params(60211, ol.RS_BSINC24, hrtf=(0.1 * (v % 9), 0.3 * (v % 11), 2.0, 0.0, 0.2)))
    tickets, known = [], {}
    for k in range(6):
        sc.mix(1024, post_process=True)
        tickets.append(sc.voice_events_async())
        if k >= 2:
            for e in sc.voice_events_wait(tickets[k - 2]):
                known[e.voice] = (e.play_state, e.has_buffer, e.position, e.position_frac)
    for t in tickets[-2:]:
        for e in sc.voice_events_wait(t):
            known[e.voice] = (e.play_state, e.has_buffer, e.position, e.position_frac)
    assert set(known) == ending                         # (voices the host started and that play on are no news)
    for v in range(V):
        st = sc.voice_state(v)
        if v in ending:
            assert st.play_state == oalgpu.VOICE_STOPPED and known[v][0] == oalgpu.VOICE_STOPPED and known[v][1] == 0, (v, known[v])
        else:
            assert st.play_state == oalgpu.VOICE_PLAYING
    # nothing changes any more: an empty report
    sc.mix(1024, post_process=True)
    sc.mix(1024, post_process=True)
    sc.voice_events_wait(sc.voice_events_async())
    sc.mix(1024, post_process=True)
    assert sc.voice_events_wait(sc.voice_events_async()) == []
    sc.close()


def test_the_first_voice_events_report_leaves_unused_slots_out(synth_mhr):
    """a context created for many more voices than it plays (BatchMixer: max_voices slots, a handful of sources): the first report
    used to list every slot nobody ever initialised -- Stopped against a snapshot seeded with another buffer index -- and with
    more than 1024 of them oalgpu_voice_events_wait failed with OALGPU_ERR_CAPACITY."""
    import oalgpu
    import oracle_lib as ol
    api = oalgpu.Api(oalgpu.MATH_FAST)
    api._mhr = open(synth_mhr, "rb").read()
    sc = api.make_scene(num_dry=4, num_real=2, hrtf=True, max_voices=3000, max_buffers=2)
    rng = np.random.default_rng(6)
    buf = sc.add_buffer(rng.uniform(-1, 1, 48000).astype(np.float32), oalgpu.FMT_FLOAT, loop_start=0, loop_end=48000)
    for v in range(5):
        sc.add_voice(buf, True, position=100 * v)
        sc.set_params(v, ol.make_voice_params(60211, ol.RS_BSINC24, hrtf=(0.1, 0.3 * v, 2.0, 0.0, 0.2)))
    sc.mix(1024, post_process=True)
    assert sc.voice_events_wait(sc.voice_events_async()) == []       # (started by the host, playing on: no news; 2995 slots never used)
    sc.close()
