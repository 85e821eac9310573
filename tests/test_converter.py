"""SampleConverter (core/converter.cpp:175-330): oalgpu_converter_* against the compiled
reference's own SampleConverter, both driven through the same ragged sequence of convert() calls -- tiny inputs
that only fill the prep samples, inputs longer than one 1024-frame chunk, outputs cut short by dst_frames so
that input is left over and fed again.  Everything a caller can observe must agree exactly: frames written,
source bytes consumed, frames left, availableOut(), and the output bytes (the integer formats and, the
resamplers running in the reference's operation order, the float ones too: bit for bit)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
NP_TYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.float32]        # DevFmtType order
CASES = [
    # (src type, dst type, channels, src rate, dst rate, resampler)
    (6, 6, 1, 44100, 48000, ol.RS_BSINC24),
    (2, 6, 2, 48000, 44100, ol.RS_BSINC24),
    (6, 2, 2, 44100, 48000, ol.RS_SPLINE if hasattr(ol, "RS_SPLINE") else 2),
    (1, 3, 3, 22050, 48000, ol.RS_LINEAR),
    (4, 0, 1, 48000, 8000, ol.RS_BSINC12 if hasattr(ol, "RS_BSINC12") else 5),
    (5, 5, 4, 48000, 48000, ol.RS_LINEAR),                 # equal rates: the copy "resampler"
    (0, 4, 2, 11025, 48000, ol.RS_POINT),
    (6, 6, 6, 96000, 48000, ol.RS_BSINC48 if hasattr(ol, "RS_BSINC48") else 9),
]


def make_input(typ, frames, channels, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(frames)[:, None]
    x = 0.6 * np.sin(2 * np.pi * (0.01 + 0.003 * np.arange(channels)[None, :]) * t) + rng.uniform(-0.3, 0.3, (frames, channels))
    if typ == 6:
        return x.astype(np.float32)
    info = np.iinfo(NP_TYPES[typ])
    half = (int(info.max) - int(info.min) + 1) // 2
    v = np.round(x * (half - 1)).astype(np.int64) + (half if info.min == 0 else 0)
    return np.clip(v, info.min, info.max).astype(NP_TYPES[typ])


@pytest.mark.parametrize("case", CASES, ids=[f"{NP_TYPES[c[0]].__name__}->{NP_TYPES[c[1]].__name__}x{c[2]}_{c[3]}->{c[4]}" for c in CASES])
def test_converter_matches_reference(case):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference")
    src_t, dst_t, ch, srate, drate, rs = case
    R = ol.load("ref").L
    R.oal_converter_create.restype = C.c_void_p
    R.oal_converter_create.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
    R.oal_converter_available_out.restype = C.c_uint32
    R.oal_converter_available_out.argtypes = [C.c_void_p, C.c_uint32]
    R.oal_converter_convert.restype = C.c_uint32
    R.oal_converter_convert.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]
    R.oal_converter_destroy.argtypes = [C.c_void_p]
    G = oalgpu.lib
    G.oalgpu_converter_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
    G.oalgpu_converter_available_out.restype = C.c_uint32
    G.oalgpu_converter_available_out.argtypes = [C.c_void_p, C.c_uint32]
    G.oalgpu_converter_convert.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_void_p, C.c_uint32]
    G.oalgpu_converter_destroy.argtypes = [C.c_void_p]
    G.oalgpu_converter_destroy.restype = None

    ref = R.oal_converter_create(src_t, dst_t, ch, srate, drate, rs)
    assert ref
    gh = C.c_void_p()
    assert G.oalgpu_converter_create(0, src_t, dst_t, ch, srate, drate, rs, C.byref(gh)) == 0, G.oalgpu_last_error()
    frame_bytes = ch * np.dtype(NP_TYPES[src_t]).itemsize
    # (input frames, dst_frames) per call: prep-only crumbs, multi-chunk blocks, outputs cut short
    calls = [(3, 64), (10, 64), (40, 4096), (700, 4096), (2500, 8192), (1, 16), (3000, 500), (1024, 1024), (5, 4096), (4000, 8192)]
    data = make_input(src_t, sum(c[0] for c in calls) + 10, ch, 11 + src_t + ch)
    at, outs, sounded = 0, [], False
    for k, (nin, ndst) in enumerate(calls):
        chunk = np.ascontiguousarray(data[at:at + nin])
        at += nin
        assert R.oal_converter_available_out(ref, nin) == G.oalgpu_converter_available_out(gh, nin), k
        # feed until this block is used up (a short dst leaves input over, like a real caller's loop)
        off, left = 0, nin
        for _ in range(64):
            if left == 0:
                break
            want = np.zeros((ndst, ch), NP_TYPES[dst_t]); got = np.zeros((ndst, ch), NP_TYPES[dst_t])
            sub = np.ascontiguousarray(chunk[off:off + left])
            r_left, cons = C.c_uint32(left), C.c_uint64(0)
            nr = R.oal_converter_convert(ref, sub.ctypes.data_as(C.c_void_p), C.byref(r_left), want.ctypes.data_as(C.c_void_p), ndst, C.byref(cons))
            gp, g_left = C.c_void_p(sub.ctypes.data), C.c_uint32(left)
            ng = G.oalgpu_converter_convert(gh, C.byref(gp), C.byref(g_left), got.ctypes.data_as(C.c_void_p), ndst)
            assert ng == nr, (k, ng, nr, G.oalgpu_last_error())
            assert g_left.value == r_left.value and (gp.value - sub.ctypes.data) == cons.value, (k, g_left.value, r_left.value)
            assert np.array_equal(got[:nr].view(np.uint8), want[:nr].view(np.uint8)), (k, float(np.abs(got[:nr].astype(np.float64) - want[:nr].astype(np.float64)).max()))
            sounded = sounded or (nr > 0 and float(np.abs(want[:nr].astype(np.float64)).max()) > 0)
            off += int(cons.value) // frame_bytes
            left = r_left.value
        assert left == 0, k
    assert sounded
    R.oal_converter_destroy(ref)
    G.oalgpu_converter_destroy(gh)
