"""The forms of the HRTF voice kernel against the oracle (same scenes, same tolerances as
tests/test_gpu_parity.py): the dual-ear FIR on the matrix pipe in split half precision (the default) and as
packed fp32 VALU FMAs (OALGPU_CTX_FIR_VALU).  The variant is chosen by oalgpu_context_desc::flags."""
import numpy as np
import pytest

import oracle_lib as ol
from scenes import run_scene

pytestmark = pytest.mark.gpu

VARIANTS = {
    "wave16": (0, "VoiceWave16Kernel<4>"),          # the default: one voice per wavefront, four wavefronts per SIMD (voice_wave16.hip)
    "wave+mfma-f16x2": (256, "VoiceWaveKernel<17, 64, 0, false, true>"),     # OALGPU_CTX_WAVE_PAIRS: two voices per wavefront
    "wave+valu": (1, "VoiceWaveKernel<17, 64, 0, false>"),
}
CASES = [
    dict(fmt=ol.FMT_FLOAT, resampler=ol.RS_BSINC24, steps=[60211], n_updates=4, nvoices=24),
    dict(fmt=ol.FMT_SHORT, resampler=ol.RS_FAST_BSINC12, steps=[70000, 52000, 65536], n_updates=4, nvoices=13),
    dict(fmt=ol.FMT_FLOAT, resampler=ol.RS_SPLINE, steps=[60211, 200000], n_updates=3, nvoices=9),
    dict(fmt=ol.FMT_FLOAT, resampler=ol.RS_LINEAR, steps=[31000], n_updates=3, nvoices=5),
    dict(fmt=ol.FMT_MULAW, resampler=ol.RS_BSINC48, steps=[60211], n_updates=3, nvoices=6),
    dict(fmt=ol.FMT_INT, resampler=ol.RS_FAST_BSINC12, steps=[60211, 500000], n_updates=5, nvoices=8, nonloop=True, stop_at=2),
    dict(fmt=ol.FMT_FLOAT, resampler=ol.RS_BSINC24, steps=[60211, 48000], n_updates=4, nvoices=17, todo=700, stop_at=1),
    dict(fmt=ol.FMT_SHORT, resampler=ol.RS_BSINC24, steps=[60211], n_updates=3, nvoices=11, todo=37),
]


@pytest.mark.parametrize("variant_env", list(VARIANTS))
@pytest.mark.parametrize("case", range(len(CASES)))
def test_variant_matches_the_oracle(variant_env, case, synth_mhr):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    which = "ref" if ol.available("ref") else "port"
    oracle = ol.load(which)
    oracle.L.oal_set_simd(1)
    cfg = dict(hrtf=True, **CASES[case])
    names = []
    ref_f, ref_i = run_scene(oracle, synth_mhr, rng_seed=11 + case, **cfg)
    got_f, got_i = run_scene(oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=VARIANTS[variant_env][0]), synth_mhr, rng_seed=11 + case, kernel_names=names, **cfg)
    assert names and all(n == VARIANTS[variant_env][1] for n in names), names
    assert got_i == ref_i, "integer voice state differs from the oracle"
    err = float(np.max(np.abs(got_f.astype(np.float64) - ref_f)))
    bound = 2e-5 * float(np.max(np.abs(ref_f))) + 1e-7
    assert err <= bound, (variant_env, case, err, bound)
