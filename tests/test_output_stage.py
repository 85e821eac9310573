"""The stage behind the buses against the reference:

  * BFormatDec::process (core/bformatdec.cpp:60-95), single band (stereo, panning.cpp StereoConfig) and
    dual band (7.1, panning.cpp X71Config with its BandSplitter per dry line): EXACT contexts bit-exact,
    FAST contexts within the FAST tolerance, splitter state carried over several updates;
  * ApplyDither + Write<T> / SampleConv<T> (alc/alu.cpp:2309-2408), every DevFmtType, with and without
    dither, partial updates, silent extra channels: the PCM is BIT-EXACT against the reference's own
    output stage (run inside DeviceBase::renderSamples by the compiled bridge) in every mode;
  * BASELINE configs[1] end to end: 4096 voices -> 5 ambisonic dry lines -> 7.1 speaker feeds -> 16-bit
    interleaved PCM.
"""
import numpy as np
import pytest

import bridge_lib as bl
import oracle_lib as ol

from oalgpu import synth

STEREO = synth.stereo_decoder()[0]          # alc/panning.cpp:548-556


def x71_matrices():                          # alc/panning.cpp:609-631
    return synth.x71_decoder()


def scene_pair(mode_exact, nvoices, num_dry, num_real, seed=3):
    import oalgpu
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    api = oalgpu.Api(oalgpu.MATH_EXACT if mode_exact else oalgpu.MATH_FAST)
    rng = np.random.default_rng(seed)
    data = rng.uniform(-1, 1, 9000).astype(np.float32)
    gains = rng.uniform(0.05, 0.3, (nvoices, num_dry))

    def build(lib, **kw):
        sc = lib.make_scene(num_dry=num_dry, num_real=num_real, hrtf=False, **kw)
        b = sc.add_buffer(data, ol.FMT_FLOAT, loop_start=0, loop_end=9000)
        for v in range(nvoices):
            sc.add_voice(b, looping=True, position=(v * 701) % 8000, frac=0)
            sc.set_params(v, ol.make_voice_params(60211, ol.RS_BSINC24, dry_gains=gains[v],
                                                  direct_filter=ol.default_filter(active=v % 2, gain_hf=0.4)))
        return sc
    return L, build(api, max_voices=nvoices), build(L)


@pytest.mark.gpu
@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("layout", ["stereo single band", "7.1 dual band"])
def test_bformat_decoder_matches_the_reference(exact, layout):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference")
    if layout.startswith("stereo"):
        nd, nr, hf, lf = 3, 2, STEREO, None
    else:
        nd, nr = 5, 8
        hf, lf = x71_matrices()
    # one voice: an EXACT context's dry lines are then bit-identical to the reference's, so the decoder is
    # compared on identical inputs
    L, gsc, osc = scene_pair(exact, 1 if exact else 7, nd, nr)
    gsc.set_bformat_decoder(hf, lf)
    odec = ol.BFormatDec(L, nd, hf, lf)
    for k, n in enumerate((1024, 1024, 700, 1024)):
        gsc.mix(n, post_process=True)
        osc.mix(n, post_process=False)
        lines = osc.dry()
        want = np.ascontiguousarray(lines[nd:])
        odec.process(want, lines[:nd], n)
        got = gsc.dry()[nd:]
        if exact:
            assert np.array_equal(got[:, :n].view(np.uint32), want[:, :n].view(np.uint32)), (k, float(np.abs(got - want).max()))
        else:
            err = float(np.abs(got[:, :n].astype(np.float64) - want[:, :n]).max())
            assert err <= 2e-5 * float(np.abs(want).max()) + 1e-7, (k, err)
        assert np.abs(want).max() > 1e-3
    odec.close(); gsc.close(); osc.close()


def sample_conv(lines, fmt, frames, frame_step):
    """SampleConv<T> + Write<T> restated (alu.cpp:2335-2390); pinned against the reference below."""
    nl = min(lines.shape[0], frame_step)
    x = np.zeros((frames, frame_step), np.float32)
    x[:, :nl] = lines[:nl, :frames].T
    if fmt == 6:
        return x.ravel()
    scale, lo, hi = {0: (128.0, -128.0, 127.0), 2: (32768.0, -32768.0, 32767.0), 4: (2147483648.0, -2147483648.0, 2147483520.0)}[fmt & ~1]
    v = np.rint(np.clip((x * np.float32(scale)).astype(np.float32), np.float32(lo), np.float32(hi))).astype(np.int64)
    if fmt & 1:
        v = v + int(scale)
    return v.astype([np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32][fmt]).ravel()


@pytest.mark.gpu
def test_output_pcm_is_bit_exact():
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    if not (ol.available("ref") and bl.available()):
        pytest.skip("needs the compiled reference and its bridge")
    L, gsc, osc = scene_pair(True, 1, 3, 2)
    gsc.set_bformat_decoder(STEREO * 6.0)            # loud enough to clip some samples in the integer formats
    odec = ol.BFormatDec(L, 3, STEREO * 6.0)
    bridge = bl.Bridge(bl.MODE_CPU)
    bl.build_config1(bridge, nsources=1)
    seed = 22222
    case = 0
    for fmt in range(7):
        for depth in (0.0, 32768.0 if fmt in (2, 3) else (128.0 if fmt < 2 else 8388608.0)):
            for n, step in ((1024, 2), (700, 3)):
                gsc.mix(n, post_process=True)
                osc.mix(n, post_process=False)
                lines = osc.dry()
                real = np.ascontiguousarray(lines[3:])
                odec.process(real, lines[:3], n)
                want, seed_after = bridge.render_lines(real, fmt, depth, seed + case, n, step)
                gsc.set_output(fmt, depth, seed + case)
                got = gsc.read_output(n, step)
                assert np.array_equal(got, want), (fmt, depth, n, step, int(np.argmax(got != want)))
                if depth == 0.0:
                    assert np.array_equal(sample_conv(real, fmt, n, step), want), (fmt, n, step)     # pins the restatement
                else:
                    # the dithered lines stay behind on both sides (ApplyDither works in place): the next
                    # read-out without dither must give their plain conversion
                    gsc.set_output(6, 0.0, 1)
                    dithered = gsc.read_output(n, 2).reshape(n, 2).T
                    q = np.rint(dithered.astype(np.float64) * depth)
                    assert np.abs(q / depth - dithered).max() < 1e-6       # quantised to the dither grid
                case += 1
    assert np.abs(want.astype(np.float64)).max() > 0
    bridge.close(); odec.close(); gsc.close(); osc.close()


@pytest.mark.gpu
def test_config2_ends_in_speaker_feeds_and_pcm(synth_mhr):
    """BASELINE configs[1], the whole way: 4096 voices -> 5 dry lines -> X71 dual-band decode -> s16 PCM."""
    import os, sys
    import oalgpu
    from oalgpu import synth
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference")
    from test_gpu_baseline_configs import build_reference_scene
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    V = 4096
    api = oalgpu.Api(oalgpu.MATH_FAST)
    api._mhr = synth.synth_mhr_bytes()
    gsc, script = bench.build_scene(oalgpu, synth, api, 2, V, 0, api._mhr, 0, num_real=8)
    osc, oscript, _ = build_reference_scene(L, synth, 2, V, synth_mhr, num_real=8)
    hf, lf = x71_matrices()
    gsc.set_bformat_decoder(hf, lf)
    gsc.set_output(oalgpu.OUT_I16, 0.0, 1)
    odec = ol.BFormatDec(L, 5, hf, lf)
    allv = list(range(V))
    moving = [v for v in allv if script.is_moving(v)]
    for k in range(3):
        voices = allv if k == 0 else moving
        gsc.set_params_batch(voices, bench.param_array(oalgpu, script, voices, k))
        for v in voices:
            osc.set_params(v, oscript.fill(ol.VoiceParams(), v, k))
        gsc.mix(1024, post_process=True)
        osc.mix(1024, post_process=False)
        lines = osc.dry()
        want = np.ascontiguousarray(lines[5:])
        odec.process(want, lines[:5], 1024)
        got = gsc.dry()[5:]
        err = float(np.abs(got.astype(np.float64) - want).max())
        assert err <= 2e-5 * float(np.abs(want).max()) + 1e-7, (k, err)
        assert np.abs(want[[0, 1, 4, 5, 6, 7]]).max() > 0.01 and np.abs(want[[2, 3]]).max() == 0.0   # no FC / LFE feed
        pcm = gsc.read_output(1024, 8).reshape(1024, 8)
        ref_pcm = sample_conv(want, 2, 1024, 8).reshape(1024, 8)
        assert np.abs(pcm.astype(np.int32) - ref_pcm).max() <= 1          # the float feeds differ in the last bits
    odec.close(); gsc.close(); osc.close()
