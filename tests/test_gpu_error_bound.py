"""The product's OWN error at BASELINE's full size, not only its distance to the reference.

tests/test_gpu_baseline_configs.py holds the 4096-voice HRTF scene to max(2e-5, 2.5 sqrt(N IrSize / 12) 2^-23)
of the reference's maximum, on the argument that the REFERENCE's serial fp32 sum into the shared accumulator
(core/mixer/hrtfbase.h:17-89) carries that much rounding noise.  Here that argument is measured: the scene's
exact result is built from the reference itself, 16 voices at a time (a 16-voice fp32 sum is good to ~1e-7 of
its own maximum) added up in float64, and both the full-size reference run and the GPU are compared with it:

    |gpu - truth| <= |reference - truth| + 1e-7 max|truth|      (maximum and RMS, every update)

i.e. the GPU is at least as close to the exact mix as the reference is -- whatever separates the two in the
full-size parity test is the reference's own noise.  Both data sets (synthetic, Default HRTF.mhr).

The same for BASELINE configs[1] (4096 voices into 5 dry lines, no HRTF): there the product keeps every line's sum in
the wavefront's registers (MixRowAcc, csrc/voice_wave.hip) and the reference adds one voice after the other into the
shared line (Mix_, core/mixer/mixer_c.cpp:150-186) -- one rounding per voice and sample instead of 64, so the
reference's own noise is smaller and the product's has to be as well."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
from test_gpu_baseline_configs import REAL_MHR, _oracle, build_reference_scene, set_reference_decoder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

V = 4096
CHUNK = 16
UPDATES = 3
LINE_MARGIN = 1e-7            # (of max|truth|; what the product may be worse by than the reference)


def _chunk_truth(L, synth, bufs, v0, todo):
    """voices [v0, v0 + CHUNK) of the bench scene alone on the reference: RealOut L/R per update and the carried
    accumulator after the last one (float32 results of a 16-voice mix)"""
    sc = ol.Scene(L, sample_rate=48000, num_dry=4, num_real=2, num_sends=0, num_slots=0, wet_channels=4, hrtf=True)
    set_reference_decoder(L, sc, synth)
    script = synth.SceneScript(3, CHUNK, v0)
    handles = {}
    for i in range(CHUNK):
        b = script.buffer_of(i, len(bufs))
        if b not in handles:
            handles[b] = sc.add_buffer(bufs[b], ol.FMT_FLOAT)
        sc.add_voice(handles[b], True, position=script.start_position(i))
    outs = []
    for k, n in enumerate(todo):
        for i in range(CHUNK):
            if k == 0 or script.is_moving(i):
                sc.set_params(i, script.fill(ol.VoiceParams(), i, k))
        sc.mix(n, post_process=True)
        outs.append(sc.dry()[4:6, :n].astype(np.float64))
    tail = sc.hrtf_accum().astype(np.float64)
    sc.close()
    return outs, tail


@pytest.mark.parametrize("data_set", ["synthetic", "Default HRTF.mhr"])
def test_gpu_is_as_close_to_the_exact_mix_as_the_reference(synth_mhr, data_set):
    import oalgpu
    from oalgpu import synth
    import bench
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    L = _oracle()
    mhr_path = synth_mhr if data_set == "synthetic" else REAL_MHR
    todo = (1024,) * UPDATES if data_set == "synthetic" else (1024, 1000, 1024)
    api = oalgpu.Api(oalgpu.MATH_FAST)
    with open(mhr_path, "rb") as f:
        mhr = f.read()
    api._mhr = mhr
    gsc, gscript = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
    osc, oscript, _ = build_reference_scene(L, synth, 3, V, mhr_path)

    # ---- the exact mix, chunk by chunk
    truth = [np.zeros((2, n)) for n in todo]
    truth_tail = None
    bufs = synth.scene_buffers(3, V)
    for v0 in range(0, V, CHUNK):
        outs, tail = _chunk_truth(L, synth, bufs, v0, todo)
        for k in range(len(todo)):
            truth[k] += outs[k]
        truth_tail = tail if truth_tail is None else truth_tail + tail

    allv = list(range(V))
    moving = [v for v in allv if gscript.is_moving(v)]
    for k, n in enumerate(todo):
        voices = allv if k == 0 else moving
        gsc.set_params_batch(voices, bench.param_array(oalgpu, gscript, voices, k))
        for v in voices:
            osc.set_params(v, oscript.fill(ol.VoiceParams(), v, k))
        gsc.mix(n, post_process=True)
        osc.mix(n, post_process=True)
        g = gsc.dry()[4:6, :n].astype(np.float64)
        r = osc.dry()[4:6, :n].astype(np.float64)
        scale = float(np.abs(truth[k]).max())
        eg, er = np.abs(g - truth[k]), np.abs(r - truth[k])
        print(f"{data_set} update {k}: |gpu - truth| max {eg.max() / scale:.2e} rms {np.sqrt((eg ** 2).mean()) / scale:.2e}; "
              f"|reference - truth| max {er.max() / scale:.2e} rms {np.sqrt((er ** 2).mean()) / scale:.2e} (of max|truth| {scale:.3e})")
        assert scale > 0.1, "the scene must sound"
        assert eg.max() <= er.max() + 1e-7 * scale, (data_set, k, eg.max() / scale, er.max() / scale)
        assert np.sqrt((eg ** 2).mean()) <= np.sqrt((er ** 2).mean()) + 1e-7 * scale, (data_set, k)
    gt, rt = gsc.hrtf_accum().astype(np.float64), osc.hrtf_accum().astype(np.float64)
    scale = float(np.abs(truth_tail).max())
    assert np.abs(gt - truth_tail).max() <= np.abs(rt - truth_tail).max() + 1e-7 * max(scale, 1e-3)
    gsc.close()
    osc.close()


def _chunk_truth_lines(L, synth, bufs, v0, todo):
    """voices [v0, v0 + CHUNK) of the config-2 scene alone on the reference: the five dry lines per update"""
    sc = ol.Scene(L, sample_rate=48000, num_dry=5, num_real=0, num_sends=0, num_slots=0, wet_channels=4, hrtf=False)
    script = synth.SceneScript(2, CHUNK, v0)
    handles = {}
    for i in range(CHUNK):
        b = script.buffer_of(i, len(bufs))
        if b not in handles:
            handles[b] = sc.add_buffer(bufs[b], ol.FMT_FLOAT)
        sc.add_voice(handles[b], True, position=script.start_position(i))
    outs = []
    for k, n in enumerate(todo):
        for i in range(CHUNK):
            if k == 0 or script.is_moving(i):
                sc.set_params(i, script.fill(ol.VoiceParams(), i, k))
        sc.mix(n, post_process=False)
        outs.append(sc.dry()[:5, :n].astype(np.float64))
    sc.close()
    return outs


def test_line_accumulators_are_as_close_to_the_exact_mix_as_the_reference(synth_mhr):
    import oalgpu
    from oalgpu import synth
    import bench
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    L = _oracle()
    todo = (1024, 1000, 1024)
    api = oalgpu.Api(oalgpu.MATH_FAST)
    with open(synth_mhr, "rb") as f:
        api._mhr = f.read()
    gsc, gscript = bench.build_scene(oalgpu, synth, api, 2, V, 0, api._mhr, 0)
    assert "DeviceLayout, 6" in gsc.voice_kernel_name(), gsc.voice_kernel_name()      # (the six-line register accumulators)
    osc, oscript, _ = build_reference_scene(L, synth, 2, V, synth_mhr)
    truth = [np.zeros((5, n)) for n in todo]
    bufs = synth.scene_buffers(2, V)
    for v0 in range(0, V, CHUNK):
        outs = _chunk_truth_lines(L, synth, bufs, v0, todo)
        for k in range(len(todo)):
            truth[k] += outs[k]
    allv = list(range(V))
    moving = [v for v in allv if gscript.is_moving(v)]
    for k, n in enumerate(todo):
        voices = allv if k == 0 else moving
        gsc.set_params_batch(voices, bench.param_array(oalgpu, gscript, voices, k))
        for v in voices:
            osc.set_params(v, oscript.fill(ol.VoiceParams(), v, k))
        gsc.mix(n, post_process=False)
        osc.mix(n, post_process=False)
        g = gsc.dry()[:5, :n].astype(np.float64)
        r = osc.dry()[:5, :n].astype(np.float64)
        scale = float(np.abs(truth[k]).max())
        eg, er = np.abs(g - truth[k]), np.abs(r - truth[k])
        print(f"config 2 update {k}: |gpu - truth| max {eg.max() / scale:.2e} rms {np.sqrt((eg ** 2).mean()) / scale:.2e}; "
              f"|reference - truth| max {er.max() / scale:.2e} rms {np.sqrt((er ** 2).mean()) / scale:.2e} (of max|truth| {scale:.3e})")
        assert scale > 0.1, "the scene must sound"
        assert eg.max() <= er.max() + LINE_MARGIN * scale, (k, eg.max() / scale, er.max() / scale)
        assert np.sqrt((eg ** 2).mean()) <= np.sqrt((er ** 2).mean()) + LINE_MARGIN * scale, k
    gsc.close()
    osc.close()
