"""Edge cases of the batched path on the GPU: tiny and odd update sizes (gain ramps longer than the
update, MixLine's `fade_len < Counter` branch), idle scenes, the widest bus the stream-row kernels
take (32 mix lines) and the first one they decline (33: VoiceMixKernel), a voice count that is not
a multiple of anything."""
import numpy as np
import pytest

import oracle_lib as ol
from scenes import run_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def libs(synth_mhr):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    which = "ref" if ol.available("ref") else "port"
    L = ol.load(which)
    L.L.oal_set_simd(1)
    return oalgpu.Api(oalgpu.MATH_FAST), L, synth_mhr


def close(a, b, what):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= 2e-5 * (np.abs(b).max() if b.size else 0.0) + 1e-7, f"{what}: {err:.3e}"


@pytest.mark.parametrize("todo", [1, 7, 63, 64, 65, 1023])
@pytest.mark.parametrize("hrtf", [False, True])
def test_odd_update_sizes(libs, todo, hrtf):
    api, L, mhr = libs
    cfg = dict(hrtf=hrtf, fmt=ol.FMT_FLOAT, resampler=ol.RS_BSINC24, steps=[60211, 90000], n_updates=5, nvoices=11,
               sends=2, todo=todo, stop_at=2)
    fa, ia = run_scene(api, mhr, rng_seed=11, **cfg)
    fb, ib = run_scene(L, mhr, rng_seed=11, **cfg)
    assert ia == ib
    close(fa, fb, f"todo {todo}")


def test_no_voices_at_all(libs):
    api, L, mhr = libs
    api.hrtf_load(mhr)
    for hrtf in (False, True):
        sc = api.make_scene(num_dry=4 if hrtf else 5, num_real=2 if hrtf else 0, num_sends=1, num_slots=1, hrtf=hrtf,
                            max_voices=16)
        if hrtf:
            sc.set_direct_hrtf(np.zeros((4, 128, 2), np.float32), [1, 1, 1, 1], 0.01, 64)
        for _ in range(3):
            sc.mix(1024, post_process=hrtf)
        assert not sc.dry().any() and not sc.wet(0).any()
        sc.close()


@pytest.mark.parametrize("form", ["rows in LDS", "stream rows"])
def test_widest_bus(libs, form):
    """28 dry + 4 wet = 32 mix lines: all 32 line accumulators of the rows-in-LDS kernel (the default), the stream-row
    kernels with S = 32 (OALGPU_CTX_STREAM_ROWS).  One more line is refused at context creation (OALGPU_ERR_CAPACITY), not
    mixed wrongly."""
    import oalgpu
    api, L, mhr = libs
    if form == "stream rows":
        api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=oalgpu.CTX_STREAM_ROWS)
    nlines = 28

    def build(lib, **kw):
        sc = lib.make_scene(num_dry=nlines, num_real=0, num_sends=1, num_slots=1, wet_channels=4, hrtf=False, **kw)
        r = np.random.default_rng(3)
        b = sc.add_buffer(r.uniform(-1, 1, 5000).astype(np.float32), ol.FMT_FLOAT, loop_start=0, loop_end=5000)
        for v in range(10):
            sc.add_voice(b, looping=True, position=(v * 433) % 4000, frac=(v * 7001) % 65536)
            snd = [(0, r.uniform(0, 0.3, 4), ol.default_filter(active=v % 2, gain_hf=0.5))]
            sc.set_params(v, ol.make_voice_params(60211, ol.RS_BSINC24, dry_gains=r.uniform(-0.2, 0.2, nlines),
                                                  direct_filter=ol.default_filter(active=(v % 3 == 0), gain_hf=0.6),
                                                  sends=snd))
        return sc

    gsc, osc = build(api, max_voices=10), build(L)
    assert ("VoiceRowsKernel" if form == "rows in LDS" else "VoiceWaveKernel") in gsc.voice_kernel_name(), gsc.voice_kernel_name()
    for k in range(3):
        gsc.mix(1000, post_process=False); osc.mix(1000, post_process=False)
        close(gsc.dry()[:, :1000], osc.dry()[:, :1000], f"dry, update {k}")
        close(gsc.wet(0)[:, :1000], osc.wet(0)[:, :1000], f"wet, update {k}")
    gsc.close(); osc.close()
    with pytest.raises(oalgpu.OalgpuError):
        api.make_scene(num_dry=29, num_real=0, num_sends=1, num_slots=1, wet_channels=4, hrtf=False)


def test_prime_voice_count(libs):
    """1009 voices: partial last workgroup, partial last line group."""
    api, L, mhr = libs
    cfg = dict(hrtf=False, fmt=ol.FMT_SHORT, resampler=ol.RS_LINEAR, steps=[60211], n_updates=2, nvoices=1009, sends=1)

    class Roomy:                      # the product's default context holds 64 voices
        @staticmethod
        def make_scene(**kw):
            return api.make_scene(max_voices=1009, **kw)

    fa, ia = run_scene(Roomy, mhr, rng_seed=5, **cfg)
    fb, ib = run_scene(L, mhr, rng_seed=5, **cfg)
    assert ia == ib
    close(fa, fb, "1009 voices")
