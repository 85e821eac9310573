"""The remaining EffectStates: chorus / flanger, distortion, autowah, vocal morpher and frequency
shifter of oalgpu_effect_* against the compiled reference's states (alc/effects/{chorus,distortion,autowah,vmorpher,
fshifter}.cpp) driven through their factories -- runs of blocks with property changes in between, ragged block
sizes, state carried across blocks (delay lines, LFO phases, filter histories, the STFT's FIFOs and overlap-add
accumulators, the gain ramps), first-order devices and devices above first order (the A-Format effects' up-sampler).

Bit for bit, except where the reference calls libm's sinf / cosf and the GPU evaluates through double precision (the
chorus' sinusoid LFO, the autowah's per-sample filter coefficients, the morpher's sinusoid LFO): those cases state
their bound."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
f32p = C.POINTER(C.c_float)

CHORUS, DISTORTION, AUTOWAH, VMORPHER, FSHIFTER, PSHIFTER = 5, 6, 7, 8, 9, 10


def _ref():
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference")
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    R = L.L
    R.oal_effect_create_ex.restype = C.c_void_p
    R.oal_effect_create_ex.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_float, C.c_uint32]
    R.oal_effect_update.argtypes = [C.c_void_p, f32p, C.c_float]
    R.oal_effect_process.argtypes = [C.c_void_p, f32p, f32p, C.c_uint32]
    R.oal_effect_destroy.argtypes = [C.c_void_p]
    return L, R


def fp(a):
    return a.ctypes.data_as(f32p)


def wet_blocks(seed, count, chans=4):
    rng = np.random.default_rng(seed)
    x = np.zeros((count, chans, 1024), np.float32)
    for u in range(count):
        if u % 4 != 3:
            x[u] = (rng.standard_normal((chans, 1024)) * 0.25).astype(np.float32)
            x[u, 1:] *= 0.5
            # something tonal under the noise, so that filters and shifters have a spectrum to work on
            t = np.arange(1024) + 1024 * u
            x[u, 0] += (0.3 * np.sin(2 * np.pi * 440.0 / 48000.0 * t)).astype(np.float32)
    return x


# per block: (props or None = no update, slot gain, samplesToDo)
SCHEDULES = {
    "chorus_triangle": (CHORUS, [([1, 90, 1.1, 0.1, 0.25, 0.016], 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 333), (None, 1.0, 1024),
                                 ([1, 0, 0.27, 1.0, -0.5, 0.002], 0.8, 1024), (None, 0.8, 1024), (None, 0.8, 1),
                                 ([1, 45, 5.0, 1.0, 0.9, 0.0001], 1.0, 1024), (None, 1.0, 700),
                                 ([1, 90, 0.0, 0.1, 0.25, 0.016], 1.0, 1024), (None, 1.0, 1024)]),
    "chorus_sinusoid": (CHORUS, [([0, -90, 3.0, 0.5, 0.5, 0.008], 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 500),
                                 ([0, 180, 10.0, 1.0, -0.9, 0.004], 0.9, 1024), (None, 0.9, 1024)]),
    "distortion": (DISTORTION, [([0.2, 0.05, 8000.0, 3600.0, 3600.0], 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 257),
                                ([0.8, 0.5, 4000.0, 1000.0, 500.0], 0.7, 1024), (None, 0.7, 1024), (None, 0.7, 3),
                                ([1.0, 1.0, 24000.0, 80.0, 100.0], 1.0, 1024), (None, 1.0, 1024)]),
    "autowah": (AUTOWAH, [([0.06, 0.06, 1000.0, 11.22], 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 400), (None, 1.0, 1024),
                          ([0.001, 0.5, 10.0, 0.5], 0.8, 1024), (None, 0.8, 1024), ([1.0, 0.0001, 2.0, 31621.0], 1.0, 1024),
                          (None, 1.0, 65)]),
    "vmorpher_triangle": (VMORPHER, [([5.0, 1, 4, 7, -5, 1], 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 300), (None, 1.0, 1024),
                                     ([20.0, 3, 2, -12, 12, 2], 0.9, 1024), (None, 0.9, 1024), (None, 0.9, 257),
                                     ([0.0, 2, 3, 0, 0, 2], 1.0, 1024), (None, 1.0, 1024),
                                     ([1.41, 0, 10, 0, 0, 1], 1.0, 1024), (None, 1.0, 1024)]),
    "vmorpher_sinusoid": (VMORPHER, [([1.41, 0, 10, 0, 0, 0], 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 600),
                                     ([8.0, 4, 1, 3, 3, 0], 0.8, 1024), (None, 0.8, 1024)]),
    "fshifter": (FSHIFTER, [([100.0, 0, 1], 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 300), (None, 1.0, 1024), (None, 1.0, 1),
                            (None, 1.0, 700), ([2500.0, 2, 0], 0.8, 1024), (None, 0.8, 1024), ([0.0, 1, 1], 1.0, 1024),
                            (None, 1.0, 1024), ([30000.0, 1, 2], 1.0, 1024), (None, 1.0, 255), (None, 1.0, 1024)]),
    "pshifter": (PSHIFTER, [([12, 0], 1.0, 1024)] + [(None, 1.0, 1024)] * 5 + [(None, 1.0, 300), ([-12, 0], 0.8, 1024)]
                 + [(None, 0.8, 1024)] * 4 + [([3, -20], 1.0, 1024)] + [(None, 1.0, 1024)] * 3 + [(None, 1.0, 1), ([0, 0], 1.0, 1024),
                                                                                                  (None, 1.0, 1024)]),
}
# compared within a bound instead of bit for bit
SINUSOID = {"chorus_sinusoid", "vmorpher_sinusoid", "autowah", "pshifter"}


def out_gain(kind, props, slot_gain):
    if kind == DISTORTION:
        return np.float32(slot_gain) * np.float32(props[1])          # slot->Gain*props.Gain, distortion.cpp:172
    return np.float32(slot_gain)


def run(name, nlines, order, wet_chans, mode, rate=48000):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    L, R = _ref()
    kind, schedule = SCHEDULES[name]
    ref = R.oal_effect_create_ex(kind, rate, nlines, 0, -1, order, 0, 400.0, wet_chans)
    assert ref
    fx = oalgpu.Effect(kind, nlines, wet_chans, rate, oalgpu.MATH_EXACT if mode == "exact" else oalgpu.MATH_FAST)
    up = None
    if kind == PSHIFTER and order > 2:
        sc, up = np.zeros(2, np.float32), np.zeros((9, 25), np.float32)
        R.oal_ambi_upmix_info2.argtypes = [C.c_uint32, C.c_int, f32p, f32p]
        R.oal_ambi_upmix_info2.restype = None
        R.oal_ambi_upmix_info2(order, 0, fp(sc), fp(up))
        fx.set_upsampler(sc, 400.0 / rate)
    elif kind != PSHIFTER and order > 1:
        sc, up, xo = L.ambi_upmix_info(order, False, rate)
        fx.set_upsampler(sc, xo)
    x = wet_blocks(60 + kind, len(schedule), wet_chans)
    if kind == AUTOWAH:
        x *= 0.5
    worst_abs, run_scale, sounded, differing, total = 0.0, 0.0, False, 0, 0
    for u, (props, gain, n) in enumerate(schedule):
        if props is not None:
            R.oal_effect_update(ref, fp(np.asarray(props, np.float32)), gain)
            g = out_gain(kind, props, gain)
            targets = np.full(max(wet_chans, 4), 0xffffffff, np.uint32)
            targets[:wet_chans] = np.arange(wet_chans)
            if up is not None:       # ComputePanGains(Dry, FirstOrderUp[c] / SecondOrderUp[c], gain) on an identity AmbiMap
                gains = ((np.float32(1.0) * up[:, :nlines]) * g).astype(np.float32)
                targets = np.full(len(up), 0xffffffff, np.uint32)
                targets[:wet_chans] = np.arange(wet_chans)
            else:
                gains = np.full(max(wet_chans, 4), g, np.float32)
            fx.update(props, targets, gains)
        want = np.zeros((nlines, 1024), np.float32); want[:, :5] = 0.125
        got = want.copy()
        R.oal_effect_process(ref, fp(np.ascontiguousarray(x[u])), fp(want), n)
        fx.process(x[u], got, n)
        assert np.array_equal(got[:, n:], want[:, n:]), "samples past samplesToDo must stay untouched"
        scale = float(np.abs(want).max())
        sounded = sounded or scale > 0.2
        if name not in SINUSOID:
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, u, float(np.abs(got - want).max()))
        else:
            worst_abs = max(worst_abs, float(np.abs(got.astype(np.float64) - want).max()))
            run_scale = max(run_scale, scale)
            differing += int((got.view(np.uint32) != want.view(np.uint32)).sum()); total += got.size
    assert sounded
    R.oal_effect_destroy(ref)
    fx.close()
    return worst_abs / max(run_scale, 1e-9), differing / max(total, 1)      # against the run's maximum


@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("name", ["chorus_triangle", "distortion", "vmorpher_triangle", "fshifter"])
def test_effect_matches_reference_bit_for_bit(name, mode):
    run(name, 4, 1, 4, mode)


def test_chorus_sinusoid_lfo():
    """sinf(offset*scale)*depth is rounded to a delay in 1/256 samples: where libm's sinf and the GPU's differ in the
    last bit AND the product sits on a rounding boundary one tap moves by 1/256 sample -- rare, local (the feedback
    path does not use the modulated delay), and small"""
    worst, frac = run("chorus_sinusoid", 4, 1, 4, "fast")
    assert worst <= 5e-3 and frac <= 0.01, (worst, frac)


def test_autowah():
    """cosf / sinf of every sample's filter frequency: a last-bit difference enters a recursive filter (Q = 5 and the
    resonance gain on top) and decays; 5e-5 of the run's maximum bounds it (measured 1e-6 .. 1.4e-5 over the rates)"""
    worst, frac = run("autowah", 4, 1, 4, "fast")
    assert worst <= 5e-5, (worst, frac)


def test_vmorpher_sinusoid_lfo():
    worst, frac = run("vmorpher_sinusoid", 4, 1, 4, "fast")
    assert worst <= 1e-6, (worst, frac)


@pytest.mark.parametrize("order", [2, 3])
@pytest.mark.parametrize("name", ["chorus_triangle", "distortion", "fshifter"])
def test_aformat_effects_on_a_higher_order_device(name, order):
    """deviceUpdate's mUpsampler: BandSplitter::processHfScale per B-Format row, then MixSamples with gains that pan
    and up-sample (chorus.cpp:393-411) -- bit for bit, and the lines above first order are fed"""
    nlines = (order + 1) ** 2
    run(name, nlines, order, 4, "fast")


def test_higher_order_lines_receive_signal():
    import oalgpu
    L, R = _ref()
    sc, up, xo = L.ambi_upmix_info(3, False)
    assert np.abs(up[:, 4:16]).max() > 0.05, "FirstOrderUp feeds lines above first order"


def test_mono_wet_bus():
    """a slot with one wet channel: only W goes into the A-Format conversion, only mChans[0] has a target"""
    run("chorus_triangle", 4, 1, 1, "fast")
    run("fshifter", 4, 1, 1, "fast")


def test_autowah_and_morpher_follow_the_wet_channel_count():
    run("vmorpher_triangle", 4, 1, 3, "fast")
    worst, _ = run("autowah", 4, 1, 2, "fast")
    assert worst <= 5e-5


@pytest.mark.parametrize("wet_chans,order", [(4, 1), (1, 1), (9, 2), (9, 3)], ids=["first_order", "mono", "second_order", "upsampled_to_third"])
def test_pitch_shifter(wet_chans, order):
    """The phase vocoder against the reference's (pffft transforms, libm atan2f / hypotf / sinf / cosf): octave up, octave
    down (several source bins per target bin: the dominant-magnitude choice), a detuned interval, unison -- 20 blocks,
    160 hops with the synthesis phases accumulating throughout.  Bound: 1e-4 of the run's maximum."""
    nlines = (order + 1) ** 2 if order > 1 else 4
    nlines = max(nlines, 9) if wet_chans == 9 else nlines
    worst, _ = run("pshifter", nlines, order, wet_chans, "fast")
    print("pitch shifter worst relative error", worst)
    assert worst <= 1e-4, worst


def test_chorus_and_pitch_shifter_on_context_slots(synth_mhr):
    """two slots of a scene, a chorus on one and a pitch shifter on the other: oalgpu_mix_update runs them between the
    reduction and the post-process from the slots' wet buses into the dry lines (the path every oalgpu_effect kind
    shares); the reference side: the oracle scene's wet buses through the compiled ChorusState / PshifterState"""
    import oalgpu
    L, R = _ref()
    api = oalgpu.Api(oalgpu.MATH_FAST)
    api.hrtf_load(synth_mhr)
    L.hrtf_load(synth_mhr)

    def build(lib):
        sc = lib.make_scene(num_dry=4, num_real=2, num_sends=2, num_slots=2, wet_channels=4, hrtf=True,
                            **({"max_voices": 8} if lib is api else {}))
        cc = np.zeros((4, 128, 2), np.float32); cc[:, :64] = np.random.default_rng(5).uniform(-0.2, 0.2, (4, 64, 2))
        sc.set_direct_hrtf(cc, [1.0, 0.8, 0.8, 0.8], 400.0 / 48000.0, 64)
        buf = sc.add_buffer(np.random.default_rng(6).uniform(-1, 1, 6000).astype(np.float32), ol.FMT_FLOAT)
        for v in range(6):
            sc.add_voice(buf, True, position=v * 700)
            r = np.random.default_rng(50 + v)
            sc.set_params(v, ol.make_voice_params(60211, ol.RS_BSINC24, hrtf=(0.2 * v, 0.9 * v, 2.0, 0.0, 0.1),
                                                  sends=[(0, r.uniform(0.1, 0.4, 4), None), (1, r.uniform(0.1, 0.4, 4), None)]))
        return sc

    chorus_props, shifter_props = [1, 90, 1.1, 0.1, 0.25, 0.016], [7, 0]
    gsc = build(api)
    fx = [oalgpu.Effect(CHORUS, 4, 4, 48000, oalgpu.MATH_FAST), oalgpu.Effect(PSHIFTER, 4, 4, 48000, oalgpu.MATH_FAST)]
    for e, props, slot in zip(fx, (chorus_props, shifter_props), (0, 1)):
        e.update(props, np.arange(4, dtype=np.uint32), np.full(4, 1.0, np.float32))
        gsc.set_slot_effect(slot, e)
    osc = build(L)
    refs = [R.oal_effect_create_ex(CHORUS, 48000, 4, 0, -1, 1, 0, 400.0, 4), R.oal_effect_create_ex(PSHIFTER, 48000, 4, 0, -1, 1, 0, 400.0, 4)]
    for r, props in zip(refs, (chorus_props, shifter_props)):
        R.oal_effect_update(r, fp(np.asarray(props, np.float32)), 1.0)
    for k in range(10):                     # the shifter's FIFO needs 1024 - 128 samples before anything comes out
        gsc.mix(1024, post_process=True)
        osc.mix(1024, post_process=False)
        dry = osc.dry_view()
        lines = np.ascontiguousarray(dry[:4])
        for slot, r in enumerate(refs):     # slots in order, each adding to the dry lines (alu.cpp:2209-2257)
            R.oal_effect_process(r, fp(np.ascontiguousarray(osc.wet(slot)[:4])), fp(lines), 1024)
        dry[:4] = lines
        osc.post_process(1024)
        a, b = gsc.dry(), osc.dry()
        assert np.abs(b).max() > 1e-3
        assert np.abs(a.astype(np.float64) - b).max() <= 1e-4 * np.abs(b).max() + 1e-7, k
    gsc.close(); osc.close()
    for e in fx:
        e.close()
    for r in refs:
        R.oal_effect_destroy(r)


@pytest.mark.parametrize("rate", [44100, 192000])
def test_other_device_rates(rate):
    """everything update() derives from the device rate -- delay-line lengths, LFO periods, filter designs, the chorus'
    history window in LDS (6150 samples at 192 kHz) -- at the rates either side of 48 kHz"""
    for name in ("chorus_triangle", "distortion", "vmorpher_triangle", "fshifter"):
        run(name, 4, 1, 4, "fast", rate)
    run("chorus_triangle", 9, 2, 4, "fast", rate)          # with the up-sampler's splitter at that rate
    worst, _ = run("autowah", 4, 1, 4, "fast", rate)
    # the filter's pole sits at cos(w0) with w0 down to 2 pi 20 Hz / rate: in float, one ulp of that cosine is a third of
    # (1 - cos w0) at 192 kHz -- the reference's own coefficients are that coarse, and so is the effect of a last-bit
    # difference between libm's cosf and the correctly rounded value
    assert worst <= (2e-4 if rate > 100000 else 5e-5), worst
    worst, _ = run("pshifter", 4, 1, 4, "fast", rate)
    assert worst <= 1e-4
