"""An HRTF device at another rate than the data set's (GetLoadedHrtf, core/hrtf.cpp:539-606) with the decoder the
reference builds for it (DirectHrtfState::build, :266-366): a 44.1 kHz and a 96 kHz context load the 48 kHz
Default HRTF.mhr, the library resamples it at load time and builds the decoder from the resampled store; voices and
post-process against the compiled reference doing the same through its own GetLoadedHrtf."""
import os
import shutil

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REAL_MHR = os.path.join(ROOT, "tests", "golden", "default_hrtf.mhr")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rate", [44100, 96000])
@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_hrtf_context_at_another_rate_than_the_data_set(tmp_path, rate, mode):
    import oalgpu
    from oalgpu import synth
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference")
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    d = tmp_path / "sets"
    d.mkdir()
    shutil.copy(REAL_MHR, d / "default_hrtf.mhr")
    info = L.hrtf_load_for_rate(str(d), rate)
    api = oalgpu.Api(oalgpu.MATH_FAST if mode == "fast" else oalgpu.MATH_EXACT)
    api.hrtf_load(REAL_MHR)
    rng = np.random.default_rng(rate)
    data = rng.uniform(-1, 1, 9000).astype(np.float32)
    nv = 14

    def params(v, k):
        r = np.random.default_rng(100 * v + k)
        return ol.make_voice_params(int(65536 * 44100 / rate) if v % 2 else 65536, ol.RS_BSINC24,
                                    hrtf=(np.arcsin(r.uniform(-1, 1)), r.uniform(-np.pi, np.pi), 2.0, 0.0, 10 ** (r.uniform(-40, -10) / 20)),
                                    direct_filter=ol.default_filter(active=v % 4 == 1, gain_hf=0.4))

    def run(lib, reference):
        sc = lib.make_scene(sample_rate=rate, num_dry=4, num_real=2, hrtf=True, **({} if reference else {"max_voices": nv}))
        if reference:
            cc, hf, irsize = L.direct_hrtf_build(info.ir_size, False, synth.AMBI_POINTS_1O, synth.AMBI_MATRIX_1O, 4, 400.0,
                                                 synth.AMBI_ORDER_HF_GAIN_1O)
            sc.set_direct_hrtf(cc, hf, 400.0 / rate, irsize)
        else:
            assert sc.hrtf_info().sample_rate == rate and sc.hrtf_info().ir_size == info.ir_size
            sc.set_direct_hrtf_from_store(synth.AMBI_POINTS_1O, synth.AMBI_MATRIX_1O, synth.AMBI_ORDER_HF_GAIN_1O, 400.0)
        b = sc.add_buffer(data, ol.FMT_FLOAT, loop_start=0, loop_end=9000)
        for v in range(nv):
            sc.add_voice(b, True, position=(v * 613) % 8000, frac=(v * 977) % 65536)
            sc.set_params(v, params(v, 0))
        out = []
        for k in range(4):
            if k:
                for v in range(0, nv, 3):
                    sc.set_params(v, params(v, k))
            sc.mix(1024, post_process=True)
            out.append(np.concatenate([sc.dry().ravel(), sc.hrtf_accum().ravel()]))
        ints = [(s.play_state, s.position, s.position_frac, tuple(s.hrtf_old_delay)) for s in (sc.voice_state(v) for v in range(nv))]
        sc.close()
        return out, ints

    want, wi = run(L, True)
    got, gi = run(api, False)
    assert gi == wi
    for k, (a, b) in enumerate(zip(got, want)):
        scale = np.abs(b).max()
        assert scale > 1e-3
        assert np.abs(a - b).max() <= 2e-5 * scale + 1e-7, (rate, mode, k, np.abs(a - b).max(), scale)
