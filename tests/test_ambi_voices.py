"""B-Format source voices (Voice::mix's ambisonic prescale, core/voice.cpp:1082-1091
-> BandSplitter::processScale, core/filters/splitter.cpp:133-161).

The compiled reference mixes one Voice with four ChannelData; the restatement and the GPU product
mix four mono voices over channel views of the interleaved buffer, each with the channel's
splitter ahead of DoFilters.  Buses must agree: bit for bit between reference and restatement
(same accumulation order), within the multi-voice tolerance on the GPU."""
import numpy as np
import pytest

import ambi_cases
import oracle_lib as ol

needs_ref = pytest.mark.skipif(not ol.available("ref"), reason="oracle/_ref not built here")


@needs_ref
@pytest.mark.parametrize("kw", [dict(), dict(sends=0, nambi=2, nmono=0), dict(resampler=ol.RS_LINEAR, seed=9),
                                dict(todo=(333, 1024, 64, 1000), seed=5)],
                         ids=["default", "ambi_only", "linear", "ragged"])
def test_port_matches_reference(kw):
    ref, port = ol.load("ref"), ol.load("port")
    ref.L.oal_set_simd(1); port.L.oal_set_simd(1)
    a = ambi_cases.run(ref, **kw)
    b = ambi_cases.run(port, **kw)
    assert np.abs(a).max() > 0.01
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())


@needs_ref
def test_prescale_is_live():
    """The splitter really is in the path: changing the LF scale of one voice changes the bus."""
    ref = ol.load("ref")
    a = ambi_cases.run(ref, nmono=0, sends=0)
    saved = ambi_cases.HF_SCALES
    try:
        ambi_cases.HF_SCALES = (1.0, 1.0, 1.0, 1.0)
        b = ambi_cases.run(ref, nmono=0, sends=0)
    finally:
        ambi_cases.HF_SCALES = saved
    assert np.abs(a - b).max() > 1e-3


# ------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["exact", "fast", "fast, a wavefront per slice"])
@pytest.mark.parametrize("kw", [dict(), dict(sends=0, nambi=2, nmono=0), dict(todo=(333, 1024, 64, 1000), seed=5)],
                         ids=["default", "ambi_only", "ragged"])
def test_gpu_matches_oracle(mode, kw):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    which = "ref" if ol.available("ref") else "port"
    L = ol.load(which)
    L.L.oal_set_simd(1)
    api = oalgpu.Api(oalgpu.MATH_EXACT if mode == "exact" else oalgpu.MATH_FAST, ctx_flags=oalgpu.CTX_SLICE_LINES if "slice" in mode else 0)
    a = ambi_cases.run(api, **kw).astype(np.float64)
    b = ambi_cases.run(L, **kw).astype(np.float64)
    # sums over several voices: the multi-voice tolerance of tests/test_gpu_parity.py
    err = np.abs(a - b).max()
    assert err <= 2e-5 * np.abs(b).max() + 1e-7, err


@pytest.mark.gpu
@pytest.mark.parametrize("channel", [0, 2])
def test_gpu_one_channel_exact_mode_is_bit_exact(channel):
    """One B-Format voice of which a single channel is audible, EXACT mode: the channel view,
    the prescale (processScale in the reference's operation order) and the mix are bit-identical
    to the reference.  (With several audible channels the partial buses of the channel voices
    are summed in a different order than the serial loop: tolerance test above.)"""
    import oalgpu
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference")
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    api = oalgpu.Api(oalgpu.MATH_EXACT)
    kw = dict(sends=0, nambi=1, nmono=0, n_updates=3, only_channel=channel)
    a = ambi_cases.run(api, **kw)
    b = ambi_cases.run(L, **kw)
    assert np.abs(b).max() > 0.01
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())
