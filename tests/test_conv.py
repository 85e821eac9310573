"""Convolution reverb (ConvolutionState, alc/effects/convolution.cpp:253-716).

CPU: the C restatement (oracle/oalport.c) against the compiled reference where it is present,
and against the committed golden lines generated from it (tests/golden/golden_conv.npz).
GPU: the HIP path through the C-ABI against the compiled reference (or, without it, the
restatement) and against the same golden lines.

Tolerance, stated once: the reference evaluates taps >= 128 through a 256-point float FFT
(pffft), the restatement through a double dot product and the GPU through its own float FFT, so
results differ by float rounding of a sum of up to 65 536 terms:
    |a - b| <= CONV_RTOL * max|reference line| + CONV_ATOL   per case."""
import os

import numpy as np
import pytest

import conv_cases
import oracle_lib as ol

CONV_RTOL = 2e-5
CONV_ATOL = 1e-7
HERE = os.path.dirname(os.path.abspath(__file__))


def close(a, b, what):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, what
    bound = CONV_RTOL * np.max(np.abs(b)) + CONV_ATOL
    err = np.max(np.abs(a - b))
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e}"


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "golden_conv.npz"))


@pytest.fixture(scope="module")
def port():
    if not ol.available("port"):
        pytest.skip("oracle/liboalport.so not built")
    return ol.load("port")


def test_port_direction_coeffs_match_golden(port, golden):
    np.testing.assert_allclose(port.direction_coeffs([0.0, 0.0, -1.0])[:4], golden["front_coeffs"][:4], rtol=1e-7)


@pytest.mark.parametrize("case", conv_cases.CASES, ids=[c[0] for c in conv_cases.CASES])
def test_port_matches_reference_golden(port, golden, case):
    front = golden["front_coeffs"]
    y = conv_cases.run_case(port.make_convolution, front, case, is_product=False)
    close(y[[0, 3]], golden[case[0]], f"port vs golden {case[0]}")
    assert np.all(y[[1, 2]] == 0.125)


@pytest.mark.skipif(not (ol.available("ref") and ol.available("port")), reason="needs both oracle libraries")
@pytest.mark.parametrize("case", conv_cases.CASES, ids=[c[0] for c in conv_cases.CASES])
def test_port_matches_compiled_reference(case):
    ref, port_ = ol.load("ref"), ol.load("port")
    front = ref.direction_coeffs([0.0, 0.0, -1.0])
    a = conv_cases.run_case(port_.make_convolution, front, case, is_product=False)
    b = conv_cases.run_case(ref.make_convolution, front, case, is_product=False)
    close(a, b, f"port vs reference {case[0]}")


def test_reference_golden_is_a_convolution(golden):
    """Sanity anchor for the fixtures themselves: after the first update's gain ramp the golden
    line 0 is slot_gain * (x * ir) to float accuracy."""
    case = conv_cases.CASES[3]
    ref = conv_cases.float_reference(case)
    y0 = golden[case[0]][0]
    n0 = case[2][0]
    np.testing.assert_allclose(y0[n0:] - 0.125, 0.4 * ref[n0:], atol=2e-5 * np.max(np.abs(ref)))


# ------------------------------------------------------------------------------------ GPU
def _gpu():
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    return oalgpu


@pytest.mark.gpu
@pytest.mark.parametrize("case", conv_cases.CASES + [conv_cases.BIG_CASE],
                         ids=[c[0] for c in conv_cases.CASES + [conv_cases.BIG_CASE]])
def test_gpu_matches_oracle(case):
    oalgpu = _gpu()
    which = "ref" if ol.available("ref") else "port"
    L = ol.load(which)
    if which == "port" and case[1] > 10000:
        pytest.skip("the O(n*taps) restatement is too slow for the 65536-tap case; golden covers it")
    front = L.direction_coeffs([0.0, 0.0, -1.0])
    a = conv_cases.run_case(lambda nl, ir: oalgpu.Convolution(nl, ir), front, case, is_product=True)
    b = conv_cases.run_case(L.make_convolution, front, case, is_product=False)
    close(a, b, f"gpu vs {L.kind} {case[0]}")


@pytest.mark.gpu
def test_gpu_matches_reference_golden(golden):
    oalgpu = _gpu()
    front = golden["front_coeffs"]
    for case in conv_cases.CASES:
        y = conv_cases.run_case(lambda nl, ir: oalgpu.Convolution(nl, ir), front, case, is_product=True)
        close(y[[0, 3]], golden[case[0]], f"gpu vs golden {case[0]}")
        assert np.all(y[[1, 2]] == 0.125)
    y = conv_cases.run_case(lambda nl, ir: oalgpu.Convolution(nl, ir), front, conv_cases.BIG_CASE, is_product=True)
    close(y[0, -1024:], golden[conv_cases.BIG_CASE[0]], "gpu vs golden 65536 taps")


@pytest.mark.gpu
def test_gpu_slot_convolution_in_scene(synth_mhr):
    """A convolution attached to an effect slot: oalgpu_mix_update feeds it channel 0 of the
    slot's wet bus and it adds into the dry lines (alc/alu.cpp:2209-2257).  Expected = the oracle
    scene's wet bus pushed through the oracle's ConvolutionState into the oracle's dry bus."""
    oalgpu = _gpu()
    from scenes import SCENES
    which = "ref" if ol.available("ref") else "port"
    L = ol.load(which)
    L.L.oal_set_simd(1)
    api = oalgpu.Api(oalgpu.MATH_FAST)
    rng = np.random.default_rng(77)
    ir = (rng.standard_normal(3000) * np.exp(-np.arange(3000) / 500.0) * 0.1).astype(np.float32)
    front = L.direction_coeffs([0.0, 0.0, -1.0])
    nlines = 5

    def build(lib):
        sc = lib.make_scene(num_dry=nlines, num_real=0, num_sends=1, num_slots=1, wet_channels=4, hrtf=False)
        r = np.random.default_rng(5)
        buf = sc.add_buffer(r.uniform(-1, 1, 9000).astype(np.float32), ol.FMT_FLOAT, loop_start=0, loop_end=9000)
        for v in range(6):
            sc.add_voice(buf, looping=True, position=(v * 977) % 8000, frac=0)
            snd = [(0, r.uniform(0.05, 0.3, 4), ol.default_filter(active=0))]
            sc.set_params(v, ol.make_voice_params(60211, ol.RS_BSINC24, dry_gains=r.uniform(0, 0.1, nlines),
                                                  direct_filter=ol.default_filter(active=0), sends=snd))
        return sc

    gsc = build(api)
    conv = oalgpu.Convolution(nlines, ir)
    conv.set_target_gains(front[:nlines] * 0.8)
    gsc.set_slot_convolution(0, conv)
    osc = build(L)
    oconv = L.make_convolution(nlines, ir)
    oconv.update(0.8)
    for k in range(4):
        n = (1024, 1000, 1024, 300)[k]
        gsc.mix(n, post_process=True)
        got = gsc.dry()
        osc.mix(n, post_process=False)
        want = osc.dry().copy()
        oconv.process(osc.wet(0)[0, :n], want)
        close(got[:, :n], want[:, :n], f"scene + slot convolution, update {k}")
    gsc.set_slot_convolution(0, None)
    conv.close(); oconv.close(); gsc.close(); osc.close()
