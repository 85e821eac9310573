"""The host's half of HrtfStore::getCoeffs (core/hrtf.cpp:192-260): oalgpu_voice_set_params and oalgpu_voice_move_async
evaluate the indices, weights and delays when they build a record (HrtfBlendFor, the same source as the device's), the GPU
does the weighted sum.  No GPU here: oalgpu_hrtf_blend_host gives the record's fields, numpy does the sum in the
reference's order, the oracle's getCoeffs is the reference.  Delays exact, coefficients to the bit."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd"))
REAL_MHR = os.path.join(HERE, "golden", "default_hrtf.mhr")


@pytest.mark.parametrize("which_set", ["synthetic", "Default HRTF.mhr"])
def test_host_blend_matches_get_coeffs(synth_mhr, which_set):
    import oalgpu
    which = "ref" if ol.available("ref") else "port"
    if not ol.available(which):
        pytest.skip("no oracle library built")
    L = ol.load(which)
    path = synth_mhr if which_set == "synthetic" else REAL_MHR
    L.hrtf_load(path)
    mhr = open(path, "rb").read()
    lib = oalgpu.lib
    vp = C.c_void_p
    lib.oalgpu_hrtf_blend_host.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, vp, C.c_size_t, vp, vp, vp, vp]
    lib.oalgpu_hrtf_parse_host.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, vp, vp, vp]
    info = (C.c_uint32 * 5)()          # oalgpu_hrtf_info: sample_rate, ir_size, num_fields, num_elevs, num_irs
    assert lib.oalgpu_hrtf_parse_host(mhr, len(mhr), 0, C.byref(info), None, None) == 0
    ir_size, num_irs = int(info[1]), int(info[4])
    coeffs = np.zeros((num_irs, 128, 2), np.float32)
    delays = np.zeros((num_irs, 2), np.uint8)
    assert lib.oalgpu_hrtf_parse_host(mhr, len(mhr), 0, C.byref(info), coeffs.ctypes.data_as(vp), delays.ctypes.data_as(vp)) == 0
    rng = np.random.default_rng(5)
    n = 300
    dirs = np.zeros((n, 4), np.float32)
    dirs[:, 0] = np.arcsin(rng.uniform(-1, 1, n))           # elevation
    dirs[:, 1] = rng.uniform(-np.pi, np.pi, n)              # azimuth
    dirs[:, 2] = rng.uniform(0.05, 3.0, n)                  # distance (several fields in the real set)
    dirs[:, 3] = np.where(rng.uniform(0, 1, n) < 0.3, rng.uniform(0, 2 * np.pi, n), 0.0)   # spread
    dirs[:4] = [[np.pi / 2, 0, 1, 0], [-np.pi / 2, 1, 1, 0], [0, np.pi, 1, 0], [0.3, -np.pi, 0.2, 6.0]]
    idx = np.zeros((n, 4), np.uint32)
    w = np.zeros((n, 4), np.float32)
    ps = np.zeros(n, np.float32)
    dl = np.zeros((n, 2), np.uint32)
    assert lib.oalgpu_hrtf_blend_host(mhr, len(mhr), 0, dirs.ctypes.data_as(vp), n, idx.ctypes.data_as(vp), w.ctypes.data_as(vp),
                                      ps.ctypes.data_as(vp), dl.ctypes.data_as(vp)) == 0
    for i in range(n):
        want_c, want_d = L.hrtf_get_coeffs(float(dirs[i, 0]), float(dirs[i, 1]), float(dirs[i, 2]), float(dirs[i, 3]))
        assert tuple(int(x) for x in want_d[:2]) == (int(dl[i, 0]), int(dl[i, 1])), (i, dirs[i], want_d, dl[i])
        # hrtf.cpp:247-259: the pass-through tap, then the four weighted HRIRs in order (float32 multiply-adds, unfused)
        got = np.zeros((128, 2), np.float32)
        got[0] = ps[i]
        for k in range(4):
            got = (coeffs[idx[i, k]] * w[i, k] + got).astype(np.float32)
        want = np.asarray(want_c, np.float32).reshape(-1, 2)
        assert np.array_equal(got[:ir_size], want[:ir_size]), (i, dirs[i], float(np.abs(got[:ir_size] - want[:ir_size]).max()))
