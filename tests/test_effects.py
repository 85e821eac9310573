"""The small EffectStates: oalgpu_effect_* against the compiled reference's EqualizerState,
ModulatorState, EchoState and DedicatedState (alc/effects/*.cpp) driven through their factories -- several blocks
of noise bursts with property changes in between, ragged block sizes, state carried across blocks (filter
histories, the echo's delay line, the carrier's phase, the gain ramps).

What update() takes from the ambisonic layer is resolved here the way the reference resolves it on a device with
identity AmbiMaps: wet channel i -> line i with the slot gain (setAmbiMixParams), the echo's taps and the dialog's
front-centre position through CalcAmbiCoeffs (the oracle's).  Both math modes: bit for bit (the sinusoid carrier:
the GPU's sinf against libm, 1e-6 of the block maximum)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
f32p = C.POINTER(C.c_float)
NLINES = 4


def _ref():
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference")
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    R = L.L
    R.oal_effect_create.restype = C.c_void_p
    R.oal_effect_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    R.oal_effect_update.argtypes = [C.c_void_p, f32p, C.c_float]
    R.oal_effect_process.argtypes = [C.c_void_p, f32p, f32p, C.c_uint32]
    R.oal_effect_targets_real.argtypes = [C.c_void_p]
    R.oal_effect_destroy.argtypes = [C.c_void_p]
    return L, R


def wet_blocks(seed, count):
    rng = np.random.default_rng(seed)
    x = np.zeros((count, 4, 1024), np.float32)
    for u in range(count):
        if u % 3 != 2:
            x[u] = (rng.standard_normal((4, 1024)) * 0.25).astype(np.float32)
            x[u, 1:] *= 0.5
    return x


def fp(a):
    return a.ctypes.data_as(f32p)


# schedules: per block (props or None = no update, slot gain, n)
EQ = [([200.0, 2.0, 500.0, 0.5, 1.0, 3000.0, 3.0, 0.7, 6000.0, 0.3], 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 300),
      ([100.0, 0.2, 800.0, 4.0, 0.3, 5000.0, 0.4, 1.0, 9000.0, 5.0], 0.7, 1024), (None, 0.7, 1024), (None, 0.7, 1)]
MOD = [([440.0, 800.0, 0], 1.0, 1024), (None, 1.0, 1024), ([1000.0, 200.0, 1], 0.8, 700), (None, 0.8, 1024),
       ([30.0, 2000.0, 2], 1.0, 1024), (None, 1.0, 1024), ([0.0, 800.0, 0], 1.0, 512)]
ECHO = [([0.1, 0.1, 0.5, 0.5, -1.0], 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 1024),
        (None, 1.0, 1024), ([0.004, 0.002, 0.2, 0.8, 0.5], 0.9, 1024), (None, 0.9, 777), (None, 0.9, 1024),
        ([0.0001, 0.0, 0.9, 0.9, 0.0], 1.0, 1024), (None, 1.0, 64)]
COMP = [([1], 1.0, 1024), (None, 1.0, 1024), (None, 1.0, 1024), ([0], 0.6, 1024), (None, 0.6, 300), ([1], 1.0, 1024), (None, 1.0, 1024)]
DED = [([0, 0.8], 1.0, 1024), (None, 1.0, 1024), ([0, 0.3], 0.5, 500), (None, 0.5, 1024)]


def targets_for(L, kind, props, slot_gain, nlines):
    """what the reference's update() resolves on a device with identity AmbiMaps"""
    if kind in (0, 1, 4):
        return np.arange(4, dtype=np.uint32), np.full(4, slot_gain, np.float32)
    if kind == 2:
        x = np.float32(props[4])
        z = np.float32(np.sqrt(np.float32(1.0) - x * x))
        g = np.zeros((2, nlines), np.float32)
        for tap, sx in enumerate((x, -x)):
            coeffs = L.direction_coeffs([-sx, 0.0, -z])        # CalcAmbiCoeffs(y = sx, z = 0, x = z)
            g[tap] = (np.float32(1.0) * coeffs[:nlines]) * np.float32(slot_gain)
        return None, g
    coeffs = L.direction_coeffs([0.0, 0.0, -1.0])              # FrontCenterCoeffs
    return None, ((np.float32(1.0) * coeffs[:nlines]) * np.float32(np.float32(slot_gain) * np.float32(props[1]))).astype(np.float32)


@pytest.mark.parametrize("mode,rate", [("exact", 48000), ("fast", 48000), ("fast", 44100), ("fast", 192000)],
                         ids=["exact", "fast", "fast_44k1", "fast_192k"])
@pytest.mark.parametrize("kind,schedule", [(0, EQ), (1, MOD), (2, ECHO), (3, DED), (4, COMP)],
                         ids=["equalizer", "modulator", "echo", "dedicated", "compressor"])
def test_effect_matches_reference(kind, schedule, mode, rate):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    L, R = _ref()
    ref = R.oal_effect_create(kind, rate, NLINES, 0, -1)
    assert ref
    fx = oalgpu.Effect(kind, NLINES, 4, rate, oalgpu.MATH_EXACT if mode == "exact" else oalgpu.MATH_FAST)
    x = wet_blocks(40 + kind, len(schedule))
    if kind == 4:
        x *= 6.0                   # the compressor's envelope moves between amplitudes 0.5 and 2
    cur_props = None
    sounded = False
    for u, (props, gain, n) in enumerate(schedule):
        if props is not None:
            cur_props = props
            R.oal_effect_update(ref, fp(np.asarray(props, np.float32)), gain)
            tg, gains = targets_for(L, kind, props, gain, NLINES)
            fx.update(None if kind == 3 else props, tg, gains)
        want = np.zeros((NLINES, 1024), np.float32); want[:, :5] = 0.125
        got = want.copy()
        R.oal_effect_process(ref, fp(x[u]), fp(want), n)
        fx.process(x[u], got, n)
        assert np.array_equal(got[:, n:], want[:, n:]), "samples past samplesToDo must stay untouched"
        err = float(np.abs(got.astype(np.float64) - want).max())
        scale = float(np.abs(want).max())
        sounded = sounded or scale > 0.2
        sin_carrier = kind == 1 and int(cur_props[2]) == 0 and cur_props[0] > 0
        if not sin_carrier:
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (u, err)
        else:
            assert err <= 1e-6 * scale + 1e-8, (u, err)
    assert sounded
    R.oal_effect_destroy(ref)
    fx.close()


def test_dedicated_dialog_on_a_real_front_centre_line():
    """a device with a FrontCenter speaker: the dialog goes to that REAL output line (dedicated.cpp:78-83)"""
    import oalgpu
    L, R = _ref()
    ref = R.oal_effect_create(3, 48000, 4, 6, 2)
    fx = oalgpu.Effect(3, 10, 4, 48000, oalgpu.MATH_EXACT)
    x = wet_blocks(9, 3)
    for u in range(3):
        if u == 0:
            R.oal_effect_update(ref, fp(np.asarray([0, 0.6], np.float32)), 0.9)
            assert R.oal_effect_targets_real(ref) == 1
            gains = np.zeros(10, np.float32)
            gains[4 + 2] = np.float32(0.9) * np.float32(0.6)         # line of the bus block: dry lines first
            fx.update(None, None, gains)
        want = np.zeros((10, 1024), np.float32); got = want.copy()
        R.oal_effect_process(ref, fp(x[u]), fp(want), 1024)
        fx.process(x[u], got, 1024)
        assert (np.abs(want[6]).max() > 0.1 or u == 2) and np.abs(want[:4]).max() == 0.0
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    R.oal_effect_destroy(ref); fx.close()


def test_effect_on_a_context_slot(synth_mhr):
    """an equalizer attached to a slot of a scene: oalgpu_mix_update runs it between the reduction and the
    post-process, from the slot's wet bus into the dry lines, like the other effects"""
    import oalgpu
    L, R = _ref()
    api = oalgpu.Api(oalgpu.MATH_FAST)
    api.hrtf_load(synth_mhr)
    L.hrtf_load(synth_mhr)

    def build(lib, with_fx):
        sc = lib.make_scene(num_dry=4, num_real=2, num_sends=1, num_slots=1, wet_channels=4, hrtf=True, **({"max_voices": 8} if lib is api else {}))
        cc = np.zeros((4, 128, 2), np.float32); cc[:, :64] = np.random.default_rng(5).uniform(-0.2, 0.2, (4, 64, 2))
        sc.set_direct_hrtf(cc, [1.0, 0.8, 0.8, 0.8], 400.0 / 48000.0, 64)
        buf = sc.add_buffer(np.random.default_rng(6).uniform(-1, 1, 6000).astype(np.float32), ol.FMT_FLOAT)
        for v in range(6):
            sc.add_voice(buf, True, position=v * 700)
            r = np.random.default_rng(50 + v)
            sc.set_params(v, ol.make_voice_params(60211, ol.RS_BSINC24, hrtf=(0.2 * v, 0.9 * v, 2.0, 0.0, 0.1),
                                                  sends=[(0, r.uniform(0.1, 0.4, 4), None)]))
        return sc

    gsc = build(api, True)
    fx = oalgpu.Effect(0, 4, 4, 48000, oalgpu.MATH_FAST)
    fx.update(EQ[0][0], np.arange(4, dtype=np.uint32), np.full(4, 1.0, np.float32))
    gsc.set_slot_effect(0, fx)
    osc = build(L, False)
    ref = R.oal_effect_create(0, 48000, 4, 0, -1)
    R.oal_effect_update(ref, fp(np.asarray(EQ[0][0], np.float32)), 1.0)
    for k in range(3):
        gsc.mix(1024, post_process=True)
        osc.mix(1024, post_process=False)
        wet = np.ascontiguousarray(osc.wet(0)[:4])
        dry = osc.dry_view()
        lines = np.ascontiguousarray(dry[:4])
        R.oal_effect_process(ref, fp(wet), fp(lines), 1024)
        dry[:4] = lines
        osc.post_process(1024)
        a, b = gsc.dry(), osc.dry()
        assert np.abs(b).max() > 1e-3
        assert np.abs(a.astype(np.float64) - b).max() <= 2e-5 * np.abs(b).max() + 1e-7, k
    gsc.close(); osc.close(); fx.close(); R.oal_effect_destroy(ref)
