"""The COMPILED reference-side binding (oracle/ref_bridge.cpp -> oracle/_ref/liboalbridge.so): BASELINE
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
