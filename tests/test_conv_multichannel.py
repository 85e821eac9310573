"""Convolution reverb with multi-channel and resampled impulse responses (ConvolutionState, alc/effects/
convolution.cpp): stereo and first-order B-Format responses (one ChannelData per channel, all fed from the
slot's input; :318-471, :623-716), responses at another sample rate (PPhaseResampler, common/
polyphase_resampler.cpp; :351-362, :412-422) and UpsampleMix on devices of a higher ambisonic order (:306-316,
:489-513).

CPU: the product's host restatement of the polyphase resampler against the compiled reference's.
GPU: the HIP path through the C-ABI against the compiled reference.  What ConvolutionState::update() computes
per channel -- the panned / rotated Target gains, the HF/LF scales, the choice of UpsampleMix -- is host-side
panning (out of scope): the tests read it out of the reference (oal_conv_channel_info) and
hand the same numbers to the product.  Tolerance: tests/test_conv.py's."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
from test_conv import close

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "openal-soft_amd"))


def _ref():
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference (oracle/_ref)")
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    return L


@pytest.mark.parametrize("src,dst,n", [(44100, 48000, 3000), (48000, 44100, 3000), (96000, 48000, 5000),
                                        (22050, 48000, 777), (32000, 48000, 1), (48000, 32000, 2)])
def test_polyphase_resampler_matches_reference(src, dst, n):
    import oalgpu
    L = _ref()
    rng = np.random.default_rng(src + dst + n)
    x = rng.standard_normal(n) * np.exp(-np.arange(n) / (n / 4.0 + 1.0))
    n_out = (n * dst + src - 1) // src
    want = L.pphase_resample(src, dst, x, n_out)
    f = oalgpu.lib.oalgpu_polyphase_resample
    f.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    got = np.zeros(n_out, np.float64)
    assert f(src, dst, x.ctypes.data_as(C.c_void_p), n, got.ctypes.data_as(C.c_void_p), n_out) == 0
    assert np.abs(want).max() > 1e-3
    # same double-precision arithmetic; the reference's Bessel/sinc constants are evaluated at compile time
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()


# (name, channels, frames, ir_rate, device order, update sizes, slot gains)
CASES = [
    ("stereo_700", 2, 700, 48000, 1, [1024, 1000, 37, 1024], [0.5, 0.5, 0.9, 0.9]),
    ("stereo_44k1_3000", 2, 3000, 44100, 1, [1024, 1024, 600, 1024], [0.4] * 4),
    ("mono_96k_9000", 1, 9000, 96000, 1, [1024, 1024, 1024], [0.6, 0.6, 0.2]),
    ("bformat_2000", 4, 2000, 48000, 1, [1024, 512, 1024, 1024], [0.7, 0.7, 0.3, 0.3]),
    ("bformat_up_order2", 4, 2000, 48000, 2, [1024, 512, 1024, 91, 1024], [0.7, 0.7, 0.3, 0.3, 0.3]),
    ("bformat_up_order3_44k1", 4, 5000, 44100, 3, [1024, 1024, 1024], [0.5] * 3),
    ("stereo_65536", 2, 65536, 48000, 1, [1024] * 4, [0.3] * 4),
]


def make_ir(name, frames, channels):
    rng = np.random.default_rng(sum(map(ord, name)))
    t = np.arange(frames)[:, None]
    ir = rng.standard_normal((frames, channels)) * np.exp(-t / max(frames / 6.0, 8.0)) * 0.2
    return np.ascontiguousarray(ir.reshape(frames) if channels == 1 else ir, np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_gpu_matches_reference(case):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    L = _ref()
    name, channels, frames, rate, order, sizes, gains = case
    nlines = (order + 1) ** 2
    ir = make_ir(name, frames, channels)
    x = np.random.default_rng(7 + frames).uniform(-1, 1, sum(sizes)).astype(np.float32)
    ref = L.make_convolution(nlines, ir, 48000, rate, device_order=order)
    if channels == 4:
        ref.set_orientation([0.3, 0.1, -0.9], [0.0, 1.0, 0.1])          # a rotated B-Format response
    gpu = oalgpu.Convolution(nlines, ir, ir_rate=rate, device_rate=48000)
    pos = 0
    for k, (n, g) in enumerate(zip(sizes, gains)):
        ref.update(g)
        tg, hf, lf, upsample, xover = ref.channel_info()
        assert tg.shape[0] == channels and upsample == (channels == 4 and order > 1)
        gpu.set_channel_gains(tg[:, :nlines])
        gpu.set_upsample(hf if upsample else None, lf if upsample else None, xover)
        want = np.full((nlines, 1024), 0.125, np.float32)
        got = want.copy()
        ref.process(x[pos:pos + n], want)
        gpu.process(x[pos:pos + n], got)
        assert np.abs(want - 0.125).max() > 1e-3
        close(got[:, :n], want[:, :n], f"{name} update {k}")
        assert np.all(got[:, n:] == 0.125), "samples past samplesToDo must stay untouched"
        pos += n
    ref.close()
    gpu.close()
