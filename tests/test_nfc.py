"""Near-field control filters: DoNfcMix, core/voice.cpp:904-932, and NfcFilter,
core/filters/nfc.cpp -- the restatement against the compiled reference (bit for bit), the GPU
against the oracle."""
import numpy as np
import pytest

import nfc_cases
import oracle_lib as ol

needs_ref = pytest.mark.skipif(not ol.available("ref"), reason="oracle/_ref not built here")
CASES = [dict(), dict(order=1, sends=0, nvoices=3), dict(order=3, seed=4), dict(order=4, nvoices=4, seed=6),
         dict(order=2, periphonic=False, todo=(100, 1024, 64, 1000)), dict(nfc_every=2, seed=8)]
IDS = ["order2", "order1", "order3", "order4", "order2_2d_ragged", "mixed_nfc_and_plain"]


@needs_ref
@pytest.mark.parametrize("kw", CASES, ids=IDS)
def test_port_matches_reference(kw):
    ref, port = ol.load("ref"), ol.load("port")
    ref.L.oal_set_simd(1); port.L.oal_set_simd(1)
    a = nfc_cases.run(ref, **kw)
    b = nfc_cases.run(port, **kw)
    assert np.abs(a).max() > 0.01
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())


@needs_ref
def test_nfc_is_live():
    """The filters are in the path: the same scene without per-voice NFC differs."""
    ref = ol.load("ref")
    a = nfc_cases.run(ref, sends=0)
    b = nfc_cases.run(ref, sends=0, nfc_every=10 ** 6)
    assert np.abs(a - b).max() > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("kw", CASES, ids=IDS)
def test_gpu_matches_oracle(kw):
    """FAST mode (the resampler and the biquads differ from the reference in the last bits, the
    NFC sections run in the reference's operation order on their inputs): the multi-voice
    tolerance of tests/test_gpu_parity.py."""
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    which = "ref" if ol.available("ref") else "port"
    L = ol.load(which)
    L.L.oal_set_simd(1)
    api = oalgpu.Api(oalgpu.MATH_FAST)
    a = nfc_cases.run(api, **kw).astype(np.float64)
    b = nfc_cases.run(L, **kw).astype(np.float64)
    err = np.abs(a - b).max()
    assert err <= 2e-5 * np.abs(b).max() + 1e-7, err


@pytest.mark.gpu
@pytest.mark.parametrize("kw", CASES, ids=IDS)
def test_gpu_exact_mode_matches_oracle(kw):
    """EXACT contexts (the workgroup-per-voice-group kernel): the NFC sections run on one lane in the reference's
    operation order.  Several voices: their sums onto a line are added in a different order than the reference's
    voice loop does, hence the multi-voice tolerance; ONE voice: bit for bit."""
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    which = "ref" if ol.available("ref") else "port"
    L = ol.load(which)
    L.L.oal_set_simd(1)
    api = oalgpu.Api(oalgpu.MATH_EXACT)
    a = nfc_cases.run(api, **kw)
    b = nfc_cases.run(L, **kw)
    err = np.abs(a.astype(np.float64) - b.astype(np.float64)).max()
    assert err <= 2e-5 * np.abs(b).max() + 1e-7, err
    one = dict(kw, nvoices=1)
    a1, b1 = nfc_cases.run(api, **one), nfc_cases.run(L, **one)
    assert np.abs(b1).max() > 1e-3
    assert np.array_equal(a1.view(np.uint32), b1.view(np.uint32)), float(np.abs(a1 - b1).max())


@pytest.mark.gpu
def test_gpu_hrtf_context_refuses_nfc(synth_mhr):
    import oalgpu
    api = oalgpu.Api(oalgpu.MATH_FAST)
    api.hrtf_load(synth_mhr)
    sc = api.make_scene(num_dry=4, num_real=2, hrtf=True)
    with pytest.raises(RuntimeError):
        sc.set_nfc(0.005, [1, 3])
    sc.close()
