"""The alternative forms of the HRTF voice kernel against the oracle (same scenes, same tolerances as
tests/test_gpu_parity.py): the matrix-pipe Toeplitz FIR inside the wavefront kernel (OALGPU_FIR=mfma)
and the workgroup-per-voice kernel (OALGPU_VOICE_KERNEL=block, at three and four workgroups per CU).
The variant is chosen when a context is created (csrc/api.hip reads the environment there)."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from scenes import run_scene

pytestmark = pytest.mark.gpu

VARIANTS = {
    "wave+mfma": ({"OALGPU_FIR": "mfma"}, "VoiceWaveKernel<17, 64, 0, false, true>"),
    "block4": ({"OALGPU_VOICE_KERNEL": "block", "OALGPU_BLOCK_WAVES": "4"}, "VoiceBlockKernel"),
    "block3": ({"OALGPU_VOICE_KERNEL": "block", "OALGPU_BLOCK_WAVES": "3"}, "VoiceBlockKernel"),
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


@pytest.fixture
def variant_env(request):
    env, _ = VARIANTS[request.param]
    old = {k: os.environ.get(k) for k in ("OALGPU_FIR", "OALGPU_VOICE_KERNEL", "OALGPU_BLOCK_WAVES")}
    for k in old:
        os.environ.pop(k, None)
    os.environ.update(env)
    yield request.param
    for k, v in old.items():
        os.environ.pop(k, None)
        if v is not None:
            os.environ[k] = v


@pytest.mark.parametrize("variant_env", list(VARIANTS), indirect=True)
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
    got_f, got_i = run_scene(oalgpu.Api(oalgpu.MATH_FAST), synth_mhr, rng_seed=11 + case, kernel_names=names, **cfg)
    assert names and all(n == VARIANTS[variant_env][1] for n in names), names
    assert got_i == ref_i, "integer voice state differs from the oracle"
    err = float(np.max(np.abs(got_f.astype(np.float64) - ref_f)))
    bound = 2e-5 * float(np.max(np.abs(ref_f))) + 1e-7
    assert err <= bound, (variant_env, case, err, bound)
