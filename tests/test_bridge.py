"""The COMPILED reference-side binding (oracle/ref_bridge.cpp -> oracle/_ref/liboalbridge.so).  First BASELINE
configs[0] -- 64 mono sources, linear resampler, stereo device, no effects -- rendered through the
reference's REAL plumbing (DeviceBase::renderSamples -> ProcessContexts -> CalcVoiceParams for every source
with pending properties -> the voice loop -> BFormatDec -> Write<float>), three ways:

  CPU       Voice::mix = the reference's own code: the baseline;
  ADAPTERS  the reference's own Voice::mix (and BFormatDec) on top of adapters with the reference's
            ResamplerFunc / MixerOutFunc signatures that call oalgpu_resample / oalgpu_mix in EXACT mode:
            bit-identical to the CPU run;
  BATCH     the voice loop replaced by ONE oalgpu_mix_update per update, the voices described by the
            oalgpu_voice_params the descriptor builder fills from the Voice objects AFTER the reference's
            CalcVoiceParams: the rendered PCM within the multi-voice tolerance,
            every source's position / fraction / play state identical.

Then the headline configuration, BASELINE configs[2], the same three ways (the second half of this file): a
RenderMode::Hrtf device set up as InitHrtfPanning does on the reference's own Default HRTF.mhr, 256 bsinc24 sources
whose Hrtf.Target the reference's CalcHrtfPanning / getCoeffs computed (alc/alu.cpp:1207-1217), every 4th moving
(MixHrtfBlend), a quarter filtered, every source with a send into an effect slot that carries the reference's own
ReverbState, DeviceBase::Process(HrtfPostProcess) behind it; f32 and 16-bit buffers.
"""
import numpy as np
import pytest

import bridge_lib as bl

UPDATES = 5


def render(mode, math_mode=1, filtered=False, stop=False, todo=(1024, 1024, 700, 1024, 1024)):
    b = bl.Bridge(mode, math_mode)
    srcs = bl.build_config1(b, filtered=filtered)
    out = []
    for k, n in enumerate(todo):
        if k:
            bl.move_some(b, srcs, k)
        if stop and k == 2:
            for v in srcs[1::7]:
                b.stop_source(v)
        out.append(b.render(n))
    states = [b.source_state(v) for v in srcs]
    b.close()
    return np.concatenate(out), states


@pytest.mark.skipif(not bl.available(), reason="needs oracle/_ref/liboalbridge.so (built where /root/reference is mounted)")
def test_reference_plumbing_renders_config1_on_the_cpu():
    """No GPU involved: the harness itself -- the reference's renderSamples with its own Voice::mix."""
    a, sa = render(bl.MODE_CPU)
    b, sb = render(bl.MODE_CPU)
    assert a.shape == (sum((1024, 1024, 700, 1024, 1024)), 2)
    assert np.array_equal(a, b) and sa == sb
    assert np.abs(a).max() > 0.05 and np.abs(a[:, 0] - a[:, 1]).max() > 0.01          # it sounds, and it is stereo
    assert all(s[0] == 1 and s[3] == 60211 for s in sa)      # Playing; mStep = fastf2u(44100/48000 * 65536), alu.cpp:1685


@pytest.mark.gpu
def test_adapters_under_the_reference_voice_mix_are_bit_exact():
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    want, sw = render(bl.MODE_CPU)
    got, sg = render(bl.MODE_ADAPTERS, math_mode=oalgpu.MATH_EXACT)
    assert sg == sw
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), float(np.abs(got - want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("math_mode", ["fast", "exact"])
@pytest.mark.parametrize("scene", ["plain", "filtered+stopping"])
def test_batched_update_behind_the_reference_voice_loop(math_mode, scene):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    kw = dict(filtered=scene != "plain", stop=scene != "plain")
    want, sw = render(bl.MODE_CPU, **kw)
    got, sg = render(bl.MODE_BATCH, math_mode=oalgpu.MATH_FAST if math_mode == "fast" else oalgpu.MATH_EXACT, **kw)
    assert sg == sw, [(i, a, b) for i, (a, b) in enumerate(zip(sg, sw)) if a != b][:4]
    err = float(np.abs(got.astype(np.float64) - want).max())
    bound = 2e-5 * float(np.abs(want).max()) + 1e-7
    assert err <= bound, (err, bound)


# ---- BASELINE configs[2]: the HRTF device, sends into a reverb slot ------------------------------------------------
HRTF_TODO = (1024, 1024, 640, 1024)


def render_hrtf(mode, math_mode=1, nsources=256, i16=False, todo=HRTF_TODO, slot_gain=1.0, stop=False, restart=False, pipelined=0):
    import oracle_lib as ol
    b = bl.Bridge(mode, math_mode, hrtf=True, num_sends=1)
    slot = b.add_reverb_slot(ol.ReverbProps.make(), slot_gain)
    if pipelined:
        b.set_pipelined(pipelined)                      # (behind the slot: the mode binds the effect slots that exist when it is entered)
    srcs = bl.build_config3(b, nsources, i16=i16, slot=slot)
    out, live = [], []
    for k, n in enumerate(todo):
        if k:
            bl.move_config3(b, srcs, k, slot=slot)
        if stop and k == 1:
            for v in srcs[1::9]:
                b.stop_source(v)
        if restart and k == 3:
            # the voices stopped at update 1 faded out during it and are Stopped: they start over as other sources
            for j, v in enumerate(srcs[1::9]):
                b.restart_source(v, (j + 3) % 8, True, 1000 + 37 * j, 0.05, (1.0, 0.5 * (j % 3 - 1), -1.5), bl.RS_BSINC24, 1.0,
                                 0.5 if j % 2 else 1.0, slot, 0.5, 1.0)
        out.append(b.render(n))
        live.append(b.batch_live_voices())
    if pipelined:
        out.extend(b.drain(1024))
        assert not b.error(), b.error()                 # (no update fell back to the CPU loop)
    states = [b.source_state(v) + b.source_flags(v) for v in srcs]
    b.close()
    return np.concatenate(out), states, live


needs_bridge = pytest.mark.skipif(not bl.available(), reason="needs oracle/_ref/liboalbridge.so (built where /root/reference is mounted)")


@needs_bridge
def test_reference_plumbing_renders_the_hrtf_device_on_the_cpu():
    """No GPU involved: the reference's renderSamples on the RenderMode::Hrtf device the bridge builds, its own
    Voice::mix, ReverbState and MixDirectHrtf."""
    a, sa, _ = render_hrtf(bl.MODE_CPU, nsources=64)
    b, sb, _ = render_hrtf(bl.MODE_CPU, nsources=64)
    assert a.shape == (sum(HRTF_TODO), 2) and np.array_equal(a, b) and sa == sb
    assert np.abs(a).max() > 0.02 and np.abs(a[:, 0] - a[:, 1]).max() > 0.005
    for v, s in enumerate(sa):
        if v % 16 == 5:         # ran out of buffer inside the third update, faded out in the fourth: Stopped, no buffer
            assert s[0] == 0 and s[4:] == (0, 1, 1), (v, s)
        else:                   # Playing, mStep = fastf2u(44100/48000 * 65536), buffer / IsFading / HasHrtf
            assert s[0] == 1 and s[3] == 60211 and s[4:] == (1, 1, 1), (v, s)
    # the slot's reverb is part of the render: without the slot's gain the output differs
    dry, _, _ = render_hrtf(bl.MODE_CPU, nsources=64, slot_gain=0.0)
    assert np.abs(a - dry).max() > 1e-4


@pytest.mark.gpu
@needs_bridge
def test_pipelined_batch_mixer_with_a_send_into_an_eax_reverb_slot():
    """BatchMixer::setPipelined on the HRTF device WITH an auxiliary send: the effect slot moves behind the boundary too -- an
    oalgpu_reverb created from the slot's ReverbProps and gain, bound with oalgpu_slot_set_reverb -- so that what the reverb adds to
    the dry lines is added in front of the device's post-process, on the GPU; the render is the reference's (its Voice::mix, its
    ReverbState, its MixDirectHrtf), two updates late."""
    import oalgpu
    todo = (1024,) * 7
    want, sw, _ = render_hrtf(bl.MODE_CPU, todo=todo, stop=True)
    got, sg, _ = render_hrtf(bl.MODE_BATCH, math_mode=oalgpu.MATH_FAST, todo=todo, stop=True, pipelined=2)
    assert got.shape[0] == want.shape[0] + 2048 and not got[:2048].any()
    scale = float(np.abs(want).max())
    err = float(np.abs(got[2048:].astype(np.float64) - want).max())
    assert err <= 4e-5 * scale + 1e-7, (err, scale)     # (the FAST reverb's block scans against the reference's serial filters)
    assert [s[0] for s in sg] == [s[0] for s in sw]
    # the reverb is part of the pipelined render: the same scene with the slot's gain at zero differs
    dry, _, _ = render_hrtf(bl.MODE_CPU, todo=todo, stop=True, slot_gain=0.0)
    assert float(np.abs(dry - want).max()) > 1e-4


@pytest.mark.gpu
@needs_bridge
@pytest.mark.parametrize("i16", [False, True])
def test_hrtf_adapters_under_the_reference_voice_mix_are_bit_exact(i16):
    """EXACT-mode per-call kernels behind the reference's function-pointer surface, under the reference's own
    Voice::mix on the HRTF device: ResamplerFunc (bsinc24), HrtfMixerFunc / HrtfMixerBlendFunc (DoHrtfMix) and
    MixerOutFunc (the sends, and the ReverbState's own mix-outs) all run on the GPU -- and the render equals the
    pure CPU render bit for bit."""
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    want, sw, _ = render_hrtf(bl.MODE_CPU, i16=i16)
    before = bl.adapter_calls()
    got, sg, _ = render_hrtf(bl.MODE_ADAPTERS, math_mode=oalgpu.MATH_EXACT, i16=i16)
    calls = [a - b for a, b in zip(bl.adapter_calls(), before)]
    assert sg == sw
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), float(np.abs(got - want).max())
    # every kind of adapter ran: 256 resamples per update, MixHrtf for every voice, MixHrtfBlend for every fading voice
    assert calls[0] >= 256 * len(HRTF_TODO) and calls[1] > 0 and calls[2] >= 256 * (len(HRTF_TODO) - 1) and calls[3] >= 256, calls


@pytest.mark.gpu
@needs_bridge
@pytest.mark.parametrize("math_mode", ["fast", "exact"])
@pytest.mark.parametrize("i16", [False, True])
def test_hrtf_batched_update_behind_the_reference_voice_loop(math_mode, i16):
    """ONE oalgpu_mix_update per update behind the reference's voice loop on the HRTF device: the descriptor builder of
    include/oalgpu_openal.hpp hands over mStep, the filter targets, Hrtf.Target (coefficients, delays, gain) and the
    sends' slots, filters and gains from the Voice objects after the reference's CalcVoiceParams; the HRTF
    accumulator and the slot's wet bus join the reference's buffers, whose ReverbState and MixDirectHrtf finish the
    update.  Sources stop, and their pooled Voice objects start over as other sources."""
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    kw = dict(i16=i16, stop=True, restart=True, todo=HRTF_TODO + (1024,))
    want, sw, _ = render_hrtf(bl.MODE_CPU, **kw)
    got, sg, live = render_hrtf(bl.MODE_BATCH, math_mode=oalgpu.MATH_FAST if math_mode == "fast" else oalgpu.MATH_EXACT, **kw)
    assert sg == sw, [(i, a, b) for i, (a, b) in enumerate(zip(sg, sw)) if a != b][:4]
    err = float(np.abs(got.astype(np.float64) - want).max())
    bound = 2e-5 * float(np.abs(want).max()) + 1e-7
    assert err <= bound, (err, bound)
    # stopped voices gave their device-side slots back, restarted ones took slots again: never more than the sources
    stopped = set(range(1, 256, 9))
    ended = {v for v in range(256) if v % 16 == 5}          # ran out of buffer in the third update, Stopped after the fourth
    assert live[0] == 256 and live[1] == 256 - len(stopped) and live[-1] == 256 - len(ended - stopped), live


# ---- the voice kinds the library has, behind the binding it ships: multi-channel static sources, streaming sources on growing
# queues, delayed starts; buffers whose storage is replaced (VERDICT r4 "missing" 2 and 3) ------------------------------------
def render_kinds(mode, kind, math_mode=1, hrtf=False, track=False, pipelined=0):
    """five updates of a small scene with one source of the kind under test among ordinary mono sources, through the
    reference's renderSamples; returns (output, per-source state incl. which buffer it plays)"""
    rng = np.random.default_rng(0xC0FFEE)
    b = bl.Bridge(mode, math_mode, hrtf=hrtf, num_sends=0)
    if pipelined:
        b.set_pipelined(pipelined)
    if track:
        b.track_changes(True)
    mono = [b.add_buffer(rng.uniform(-1, 1, 20000).astype(np.float32)) for _ in range(3)]
    srcs = [b.add_source(mono[v % 3], True, 997 * v, 0.2, (float(v - 2), 0.0, -2.0), resampler=bl.RS_BSINC24 if hrtf else bl.RS_LINEAR,
                         gain_hf=0.5 if v == 1 else 1.0) for v in range(5)]
    special, extra = None, {}
    if kind == "stereo":
        st = rng.uniform(-1, 1, (9000, 2)).astype(np.float32)
        special = b.add_source_stereo(b.add_buffer_interleaved(st, 2), False, 100, 0.3, (0.5, 0.0, -1.0), resampler=bl.RS_LINEAR)
    elif kind == "queue":
        parts = [b.add_buffer(rng.uniform(-1, 1, n).astype(np.float32)) for n in (1500, 700, 2600, 900)]
        extra["parts"] = parts
        extra["late"] = b.add_buffer(rng.uniform(-1, 1, 3000).astype(np.float32))
        special = b.add_source_queue(parts, False, 0.3, (-0.5, 0.2, -1.5), resampler=bl.RS_LINEAR)
    elif kind == "callback":
        # 3500 frames at pitch 1.3 from 44.1 kHz: the function comes up short in the third update, the source ends in it
        special = b.add_source_callback(3500, 7, 0.3, (0.5, 0.2, -1.0), resampler=bl.RS_BSINC24 if hrtf else bl.RS_SPLINE, pitch=1.3, gain_hf=0.7)
        extra["second"] = b.add_source_callback(100000, 3, 0.2, (-1.0, 0.0, -1.5), resampler=bl.RS_LINEAR)     # (plays on)
    elif kind == "delay":
        special = b.add_source(mono[0], True, 4321, 0.3, (1.0, 0.5, -1.0), resampler=bl.RS_LINEAR)
        b.set_start_delay(special, 1024 + 300)             # starts 300 samples into the SECOND update
    out, cur = [], []
    for k in range(5):
        if k:
            for v in srcs[::2]:
                b.update_source(v, 0.2 + 0.01 * k, (float(v - 2) * (1.0 - 0.1 * k), 0.1 * k, -2.0),
                                resampler=bl.RS_BSINC24 if hrtf else bl.RS_LINEAR, gain_hf=0.5 if v == 1 else 1.0)
        if kind == "queue" and k == 2:
            b.queue_buffer(extra["parts"][-1], extra["late"])      # alSourceQueueBuffers while the source plays
        out.append(b.render(1024))
        cur.append(b.source_buffer(special) if special is not None else -1)
    if pipelined:
        out.extend(b.drain(1024))
    states = [b.source_state(v) + b.source_flags(v) for v in srcs + ([special] if special is not None else []) + ([extra["second"]] if "second" in extra else [])]
    if mode == bl.MODE_BATCH and not pipelined:
        assert not b.error(), b.error()                     # (no update went back to the CPU loop)
    b.close()
    return np.concatenate(out), states, cur


@pytest.mark.gpu
@needs_bridge
@pytest.mark.parametrize("hrtf", [False, True], ids=["stereo-device", "hrtf-device"])
@pytest.mark.parametrize("kind", ["stereo", "queue", "delay", "callback"])
def test_voice_kinds_behind_the_reference_voice_loop(kind, hrtf):
    """a scene that contains ONE such source used to send every update back to the CPU loop"""
    import oalgpu
    want, sw, cw = render_kinds(bl.MODE_CPU, kind, hrtf=hrtf)
    got, sg, cg = render_kinds(bl.MODE_BATCH, kind, math_mode=oalgpu.MATH_FAST, hrtf=hrtf)
    assert sg == sw, [(i, a, b) for i, (a, b) in enumerate(zip(sg, sw)) if a != b][:4]
    assert cg == cw, (cg, cw)                                # a streaming source's current buffer follows the queue
    err = float(np.abs(got.astype(np.float64) - want).max())
    bound = 2e-5 * float(np.abs(want).max()) + 1e-7
    assert err <= bound, (kind, err, bound)
    if kind == "delay":                                      # silent before its start: the first update has only the other sources
        assert np.abs(want[1024 + 300 - 8:1024 + 300 + 200]).max() > 0


# ---- B-Format sources the device's order exceeds (VoiceFlag::IsAmbisonic) and near-field-compensated voices (VoiceFlag::HasNfc): on a
# stereo device that mixes second-order 2D ambisonics, without and with a control distance (VERDICT r5 "next" 4a) ---------------------
def render_ambi2(mode, control_distance, math_mode=1, with_bformat=True):
    rng = np.random.default_rng(0xB0F)
    b = bl.Bridge(mode, math_mode, ambi2=True, control_distance=control_distance)
    mono = [b.add_buffer(rng.uniform(-1, 1, 20000).astype(np.float32)) for _ in range(3)]
    srcs = [b.add_source(mono[v % 3], True, 997 * v, 0.2, (float(v - 2), 0.0, -2.0 + 0.4 * v), resampler=bl.RS_BSINC24 if v % 2 else bl.RS_LINEAR,
                         gain_hf=0.5 if v == 1 else 1.0) for v in range(5)]
    if with_bformat:
        bf = rng.uniform(-1, 1, (12000, 3)).astype(np.float32)
        srcs.append(b.add_source_bformat2d(b.add_buffer_interleaved(bf, 3), True, 100, 0.3, (0.0, 0.0, 0.0), resampler=bl.RS_LINEAR))
        srcs.append(b.add_source_bformat2d(b.add_buffer_interleaved(bf[::-1].copy(), 3), False, 5000, 0.25, (0.3, 0.0, -1.0),
                                           resampler=bl.RS_BSINC24, pitch=1.1, gain_hf=0.6))
    out = []
    for k in range(6):
        if k:
            for v in srcs[:5:2]:                       # (a moved source: another distance, another w0)
                b.update_source(v, 0.2 + 0.01 * k, (float(v - 2) * (1.0 - 0.1 * k), 0.1 * k, -2.0 + 0.3 * k),
                                resampler=bl.RS_LINEAR, gain_hf=1.0)
        if k == 4:
            b.stop_source(srcs[1])
        out.append(b.render(1024))
    states = [b.source_state(v) + b.source_flags(v) for v in srcs]
    err = b.error()
    b.close()
    return np.concatenate(out), states, err


@pytest.mark.gpu
@needs_bridge
@pytest.mark.parametrize("control_distance", [0.0, 1.5], ids=["no-nfc", "nfc"])
def test_bformat_and_near_field_voices_behind_the_reference_voice_loop(control_distance):
    """first-order B-Format sources on the second-order device (a device voice per channel with Voice::prepare's HF / LF scales) and
    -- with a control distance -- every voice near-field compensated (oalgpu_context_set_nfc, oalgpu_voice_set_nfc with the w0
    recovered from the voice's NFCtrlFilter): the update stays behind the boundary and matches the reference's own loop"""
    import oalgpu
    want, sw, _ = render_ambi2(bl.MODE_CPU, control_distance)
    got, sg, err = render_ambi2(bl.MODE_BATCH, control_distance, math_mode=oalgpu.MATH_FAST)
    assert not err, err                                 # (no update went back to the CPU loop)
    assert sg == sw, [(i, a, b) for i, (a, b) in enumerate(zip(sg, sw)) if a != b][:4]
    e = float(np.abs(got.astype(np.float64) - want).max())
    bound = 2e-5 * float(np.abs(want).max()) + 1e-7
    assert e <= bound, (control_distance, e, bound)
    # the kinds are audible: the same scene without its B-Format sources differs, and so does the one without near-field control
    other, _, _ = render_ambi2(bl.MODE_CPU, control_distance, with_bformat=False)
    assert float(np.abs(other - want).max()) > 1e-3
    if control_distance:
        plain, _, _ = render_ambi2(bl.MODE_CPU, 0.0)
        assert float(np.abs(plain - want).max()) > 1e-4


@pytest.mark.gpu
@needs_bridge
def test_changed_voices_only_with_the_parameter_hook():
    """BatchMixer::trackChanges: told which voices CalcSourceParams recomputed, the update hands over those voices' parameters
    only -- the result is the one with every voice compared"""
    import oalgpu
    want, sw, _ = render_kinds(bl.MODE_CPU, "none", hrtf=True)
    a, sa, _ = render_kinds(bl.MODE_BATCH, "none", math_mode=oalgpu.MATH_FAST, hrtf=True)
    t, st, _ = render_kinds(bl.MODE_BATCH, "none", math_mode=oalgpu.MATH_FAST, hrtf=True, track=True)
    assert sa == sw and st == sw
    assert np.array_equal(a.view(np.uint32), t.view(np.uint32))
    assert float(np.abs(a.astype(np.float64) - want).max()) <= 2e-5 * float(np.abs(want).max()) + 1e-7


@pytest.mark.gpu
@needs_bridge
@pytest.mark.parametrize("case", ["same-length+forget", "other-length"])
def test_a_buffer_freed_and_reallocated_at_the_same_address_plays_the_new_samples(case):
    """core/buffer_storage.h:47-77: buffer storage is freed and reused.  A source plays buffer A to its end; A's storage then holds
    other samples (the same VoiceBufferItem, the same address); a new source on it must play THOSE.  With the same length nothing
    about the item tells -- the maintainer's forgetBuffer hook (alDeleteBuffers / alBufferData) does; another length is noticed by
    the mixer itself."""
    import oalgpu
    rng = np.random.default_rng(5)
    first = rng.uniform(-1, 1, 3000).astype(np.float32)
    second = rng.uniform(-1, 1, 3000 if case.startswith("same") else 2500).astype(np.float32)

    def run(mode):
        b = bl.Bridge(mode, oalgpu.MATH_FAST, hrtf=False, num_sends=0)
        keep = b.add_source(b.add_buffer(rng.uniform(-1, 1, 30000).astype(np.float32) * 0 + 0.01), True, 0, 0.1, (0.0, 0.0, -1.0))
        buf = b.add_buffer(first)
        s1 = b.add_source(buf, False, 0, 0.5, (1.0, 0.0, -1.0))
        out = [b.render(1024) for _ in range(5)]            # s1 runs out of buffer, fades, stops
        assert b.source_state(s1)[0] == 0
        b.replace_buffer(buf, second, forget=case.startswith("same"))
        b.restart_source(s1, buf, False, 0, 0.5, (1.0, 0.0, -1.0), bl.RS_LINEAR, 1.0, 1.0, -1, 1.0, 1.0)
        out += [b.render(1024) for _ in range(2)]
        live = b.batch_live_buffers() if mode == bl.MODE_BATCH else 0
        b.close()
        return np.concatenate(out), live

    want, _ = run(bl.MODE_CPU)
    got, live = run(bl.MODE_BATCH)
    assert float(np.abs(got.astype(np.float64) - want).max()) <= 2e-5 * float(np.abs(want).max()) + 1e-7
    assert np.abs(want[5 * 1024:]).max() > 0.05             # the second run sounded
    assert live == 2                                        # the replaced copy was given up, not kept beside the new one


@pytest.mark.gpu
@needs_bridge
def test_hrtf_batched_update_at_the_headline_size():
    """BASELINE configs[2]'s size through the binding that is shipped: 4096 HRTF sources behind the reference's own
    renderSamples / ProcessContexts / CalcVoiceParams in BATCH mode (with the parameter hook, as tools/bridge_period.py times
    it) against the reference's own Voice::mix: four updates, every 4th source moving, a quarter filtered, a send into the
    reference's ReverbState; integer state exact."""
    import oalgpu
    kw = dict(nsources=4096, todo=(1024, 1024, 1024, 1024))
    want, sw, _ = render_hrtf(bl.MODE_CPU, **kw)
    got, sg, live = render_hrtf(bl.MODE_BATCH, math_mode=oalgpu.MATH_FAST, **kw)
    assert sg == sw, [(i, a, b) for i, (a, b) in enumerate(zip(sg, sw)) if a != b][:4]
    err = float(np.abs(got.astype(np.float64) - want).max())
    # 4e-5 of the maximum: twice the 256-source bound -- both sides sum sixteen times as many voices in fp32 (the reference
    # serially into HrtfAccumData, the GPU per workgroup and then across workgroups), whose rounding noise grows with the root
    # of the count; measured 2.05e-5.  (tests/test_gpu_error_bound.py bounds the product against an f64 mix at this size.)
    bound = 4e-5 * float(np.abs(want).max()) + 1e-7
    assert err <= bound, (err, bound)
    assert live[0] == 4096


# ---- the pipelined mode of the batch mixer (include/oalgpu_openal.hpp) --------------------------------------
def render_hrtf_direct(mode, nsources, updates, pipelined=0, stop=True, track=False, restart=False, hook=False):
    """The HRTF device without auxiliary sends: config-3-shaped sources, every 4th moving, one in sixteen running out of buffer in
    the third update, a ninth told to stop in the second.  pipelined: the batch mixer's pipelined mode with that depth; the
    outstanding updates are drained at the end.  -> ([updates (+ depth)][1024][2], play states)"""
    b = bl.Bridge(mode, 1, hrtf=True, num_sends=0)
    if pipelined:
        b.set_pipelined(pipelined)
    if track:
        b.track_changes(True)
    if hook:
        b.hook_alu(True)            # the hooks inside alc/alu.cpp: directions instead of blended responses, changed voices named by CalcVoiceParams
    srcs = bl.build_config3(b, nsources, slot=-1)
    out = []
    for k in range(updates):
        if k:
            bl.move_config3(b, srcs, k, slot=-1)
        if stop and k == 1:
            for v in srcs[1::9]:
                b.stop_source(v)
        if restart and k == 5:
            # the voices stopped at update 1 faded out during it and are Stopped: their Voice objects start over as other sources
            # (on the GPU side: other device slots, or the same ones again -- a late report about the old voice must not reach the new)
            for j, v in enumerate(srcs[1::9]):
                b.restart_source(v, (j + 3) % 8, True, 1000 + 37 * j, 0.05, (1.0, 0.5 * (j % 3 - 1), -1.5), bl.RS_BSINC24, 1.0,
                                 0.5 if j % 2 else 1.0, -1, 0.5, 1.0)
        out.append(b.render(1024))
    if pipelined:
        out.extend(b.drain(1024))
    states = [b.source_state(v)[0] for v in srcs]
    live = b.batch_live_voices() if mode == bl.MODE_BATCH else 0
    if hook:
        # every recomputed voice behind the first update went through the hook: a quarter of the sources per update
        assert b.hooked_directions() >= (updates - 2) * (nsources // 4), b.hooked_directions()
    b.close()
    return np.stack(out), states, live


@pytest.mark.gpu
@needs_bridge
@pytest.mark.parametrize("nsources,restart", [(256, True), (4096, False)])
def test_the_getcoeffs_hook_hands_over_directions_instead_of_responses(nsources, restart):
    """SURVEY.md 8 f1 behind the shipped binding (include/oalgpu_openal_hooks.hpp): the bridge library is built with alc/alu.cpp plus
    the binding's four lines (oracle/_ref/alu_hooked.cpp) -- CalcPanningAndFilters hands the batch mixer the DIRECTION of every
    voice it recomputes and never blends a response, the device context evaluates HrtfStore::getCoeffs from 24-byte move records
    -- against the reference's own render (its getCoeffs, its Voice::mix, its MixDirectHrtf) of the same scene: a quarter of the
    sources moving in every update, two updates late, integer state exact."""
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    U, D = 11, 2
    want, sw, _ = render_hrtf_direct(bl.MODE_CPU, nsources, U, restart=restart)
    got, sg, live = render_hrtf_direct(bl.MODE_BATCH, nsources, U, pipelined=D, hook=True, restart=restart)
    assert got.shape[0] == U + D and not got[:D].any()
    scale = float(np.abs(want).max())
    assert scale > 0.02
    tol = (4e-5 if nsources > 1000 else 2e-5) * scale + 1e-7
    for u in range(U):
        err = float(np.abs(got[u + D].astype(np.float64) - want[u]).max())
        assert err <= tol, (u, err, tol)
    assert sg == sw, [(i, a, b) for i, (a, b) in enumerate(zip(sg, sw)) if a != b][:6]
    assert live == sum(1 for s in sw if s == 1)


@pytest.mark.gpu
@needs_bridge
def test_the_getcoeffs_hook_in_the_synchronous_form():
    """the same hook without the pipelined mode (every update's buses read back): 256 sources, six updates"""
    import oalgpu
    U = 6
    want, sw, _ = render_hrtf_direct(bl.MODE_CPU, 256, U)
    got, sg, live = render_hrtf_direct(bl.MODE_BATCH, 256, U, hook=True)
    scale = float(np.abs(want).max())
    for u in range(U):
        err = float(np.abs(got[u].astype(np.float64) - want[u]).max())
        assert err <= 2e-5 * scale + 1e-7, (u, err)
    assert sg == sw


@pytest.mark.gpu
@needs_bridge
@pytest.mark.parametrize("nsources,track,restart", [(256, False, False), (256, True, True), (4096, True, False)])
def test_pipelined_batch_mixer_delivers_the_same_render_two_updates_late(nsources, track, restart):
    """BatchMixer::setPipelined(2): voices AND the HRTF post-process behind the boundary, an update's two output lines added to
    RealOut two updates later, voice state heard of through change reports -- against the reference's own render of the same
    scene (its Voice::mix, its MixDirectHrtf): update u of the pipelined render is update u - 2 of the reference's, the first
    two are silence, the last two come out of the drain; sources that end or are stopped reach the same play state."""
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    # (the last of the 4096 sources that run out of buffer does so in the seventh update; the pipelined side hears of it two
    # updates later and the source is Stopped the update after that)
    U, D = 11, 2
    want, sw, _ = render_hrtf_direct(bl.MODE_CPU, nsources, U, restart=restart)
    got, sg, live = render_hrtf_direct(bl.MODE_BATCH, nsources, U, pipelined=D, track=track, restart=restart)
    assert got.shape[0] == U + D and not got[:D].any()
    scale = float(np.abs(want).max())
    assert scale > 0.02
    tol = (4e-5 if nsources > 1000 else 2e-5) * scale + 1e-7        # (the bounds of the synchronous tests above)
    for u in range(U):
        err = float(np.abs(got[u + D].astype(np.float64) - want[u]).max())
        assert err <= tol, (u, err, tol)
    assert sg == sw, [(i, a, b) for i, (a, b) in enumerate(zip(sg, sw)) if a != b][:6]
    assert live == sum(1 for s in sw if s == 1)         # every source that stopped or ended gave its device slot back


@pytest.mark.gpu
@needs_bridge
@pytest.mark.parametrize("hrtf", [True, False], ids=["hrtf-device", "stereo-device"])
@pytest.mark.parametrize("kind", ["stereo", "queue", "delay"])
def test_pipelined_batch_mixer_with_the_other_voice_kinds(kind, hrtf):
    """the pipelined mode with a multi-channel source, a streaming source on a growing queue (its progress comes back in the change
    reports) and a delayed start among the mono sources: the render is the reference's, two updates late -- on the HRTF device
    (HrtfPostProcess behind the boundary) and on the stereo device (AmbiDecPostProcess: the BFormatDec speaker decode behind it,
    the real output lines coming back)"""
    import oalgpu
    want, _, _ = render_kinds(bl.MODE_CPU, kind, hrtf=hrtf)
    got, _, _ = render_kinds(bl.MODE_BATCH, kind, math_mode=oalgpu.MATH_FAST, hrtf=hrtf, track=True, pipelined=2)
    assert got.shape[0] == want.shape[0] + 2 * 1024 and not got[:2048].any()
    err = float(np.abs(got[2048:].astype(np.float64) - want).max())
    bound = 2e-5 * float(np.abs(want).max()) + 1e-7
    assert err <= bound, (kind, err, bound)


@pytest.mark.gpu
@needs_bridge
def test_leaving_the_pipelined_mode_hands_the_post_process_back():
    """BatchMixer::leavePipelined after a drain: the device has its HrtfPostProcess again and the synchronous form carries on --
    the render is the reference's but for one seam (the HRTF accumulator's tail of the last pipelined update stays on the GPU)."""
    import oalgpu
    U1, U2, n = 5, 4, 256
    want, _, _ = render_hrtf_direct(bl.MODE_CPU, n, U1 + U2, stop=False)
    b = bl.Bridge(bl.MODE_BATCH, 1, hrtf=True, num_sends=0)
    b.set_pipelined(2)
    srcs = bl.build_config3(b, n, slot=-1)
    out = []
    for k in range(U1 + U2):
        if k:
            bl.move_config3(b, srcs, k, slot=-1)
        if k == U1:
            out.extend(b.drain(1024))
            b.leave_pipelined()
        out.append(b.render(1024))
    b.close()
    got = np.stack(out)
    assert got.shape[0] == U1 + U2 + 2 and not got[:2].any()
    scale = float(np.abs(want).max())
    for u in range(U1 + U2):
        a, w = got[u + 2].astype(np.float64), want[u]
        if u == U1:
            a, w = a[128:], w[128:]                  # (the seam: HrirLength samples)
        assert float(np.abs(a - w).max()) <= 2e-5 * scale + 1e-7, u
