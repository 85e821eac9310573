"""One-time HRTF set-up the reference does on the CPU, restated in the product's host code and pinned to the compiled
reference (no GPU needed):

* DirectHrtfState::build (core/hrtf.cpp:266-366) -- the ambisonic-to-binaural decoder of the HRTF post-process from the
  virtual-speaker layout of InitHrtfPanning (alc/panning.cpp:861-1038): bit for bit;
* GetLoadedHrtf's resampling of a data set to the device's rate (core/hrtf.cpp:539-606): delays and IrSize exact, the
  HRIRs bit for bit (the polyphase sums are double; both sides narrow them to the same floats)."""
import ctypes as C
import os
import shutil

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REAL_MHR = os.path.join(ROOT, "tests", "golden", "default_hrtf.mhr")


def _ref():
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference (oracle/_ref)")
    return ol.load("ref")


def _product_build(mhr_bytes, rate, irsize, per_min, points, matrix, nchans, xover, gains):
    import oalgpu
    lib = oalgpu.lib
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
    mat = np.zeros((len(pts), 16), np.float32)
    mat[:, :np.asarray(matrix).shape[1]] = matrix
    g = np.zeros(5, np.float32)
    g[:len(gains)] = gains
    co = np.zeros((nchans, 128, 2), np.float32)
    hf = np.zeros(nchans, np.float32)
    xn, ir = C.c_float(0), C.c_uint32(0)
    f32p = C.POINTER(C.c_float)
    lib.oalgpu_hrtf_build_direct_host.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, f32p, f32p, C.c_uint32,
                                                  C.c_uint32, C.c_float, f32p, f32p, f32p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    rc = lib.oalgpu_hrtf_build_direct_host(mhr_bytes, len(mhr_bytes), rate, irsize, 1 if per_min else 0,
                                           pts.ctypes.data_as(f32p), mat.ctypes.data_as(f32p), len(pts), nchans, xover,
                                           g.ctypes.data_as(f32p), co.ctypes.data_as(f32p), hf.ctypes.data_as(f32p), C.byref(xn), C.byref(ir))
    assert rc == 0, oalgpu.lib.oalgpu_last_error()
    return co, hf, xn.value, ir.value


@pytest.mark.parametrize("data_set", ["synthetic", "Default HRTF.mhr"])
@pytest.mark.parametrize("layout", ["first order (InitHrtfPanning)", "random 14 points, 9 channels, per-HRIR delays"])
def test_direct_hrtf_build_bit_exact(synth_mhr, data_set, layout):
    from oalgpu import synth
    L = _ref()
    path = synth_mhr if data_set == "synthetic" else REAL_MHR
    info = L.hrtf_load(path)
    with open(path, "rb") as f:
        mhr = f.read()
    if layout.startswith("first"):
        pts, mat, gains, nch, per_min = synth.AMBI_POINTS_1O, synth.AMBI_MATRIX_1O, synth.AMBI_ORDER_HF_GAIN_1O, 4, False
    else:
        rng = np.random.default_rng(7)
        pts = np.stack([np.arcsin(rng.uniform(-1, 1, 14)), rng.uniform(-np.pi, np.pi, 14)], axis=1).astype(np.float32)
        mat = rng.uniform(-0.2, 0.2, (14, 9)).astype(np.float32)
        gains, nch, per_min = np.array([1.8, 1.4, 0.7, 0.0, 0.0], np.float32), 9, True
    want_co, want_hf, want_ir = L.direct_hrtf_build(info.ir_size, per_min, pts, mat, nch, 400.0, gains)
    got_co, got_hf, xn, got_ir = _product_build(mhr, 0, 0, per_min, pts, mat, nch, 400.0, gains)
    assert got_ir == want_ir and 8 <= got_ir <= 128
    assert np.array_equal(got_hf, want_hf)
    assert np.array_equal(got_co.view(np.uint32), want_co.view(np.uint32)), float(np.abs(got_co - want_co).max())
    assert np.abs(want_co).max() > 1e-3
    assert xn == np.float32(400.0 / info.sample_rate)


@pytest.mark.parametrize("rate", [44100, 96000, 32000])
def test_data_set_resampled_to_the_device_rate(tmp_path, rate):
    import oalgpu
    L = _ref()
    d = tmp_path / "sets"
    d.mkdir()
    shutil.copy(REAL_MHR, d / "default_hrtf.mhr")
    info = L.hrtf_load_for_rate(str(d), rate)
    want = L.hrtf_raw()
    assert info.sample_rate == rate
    with open(REAL_MHR, "rb") as f:
        mhr = f.read()
    gi = oalgpu.HrtfInfo()
    co = np.zeros((info.num_irs, 128, 2), np.float32)
    de = np.zeros((info.num_irs, 2), np.uint8)
    lib = oalgpu.lib
    lib.oalgpu_hrtf_parse_host.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.oalgpu_hrtf_parse_host(mhr, len(mhr), rate, C.byref(gi), co.ctypes.data, de.ctypes.data) == 0
    assert (gi.sample_rate, gi.ir_size, gi.num_irs) == (rate, info.ir_size, info.num_irs)
    assert np.array_equal(de, want["delays"])
    err = np.abs(co - want["coeffs"]).max()
    assert np.array_equal(co.view(np.uint32), want["coeffs"].view(np.uint32)), float(err)
    # and it is not the identity
    gi0 = oalgpu.HrtfInfo()
    co0 = np.zeros_like(co)
    assert lib.oalgpu_hrtf_parse_host(mhr, len(mhr), 0, C.byref(gi0), co0.ctypes.data, None) == 0
    assert gi0.sample_rate == 48000 and np.abs(co0 - co).max() > 1e-3
