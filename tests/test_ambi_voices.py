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
