"""Pins the plain-C restatement (oracle/oalport.c) against the REAL reference mixer compiled
in place from /root/reference (oracle/_ref/liboalref.so).  Everything is compared BIT-EXACT:
the port reproduces the reference's arithmetic order (SSE and C variants), FTZ/DAZ and
no-FMA rounding, so there is no tolerance anywhere in this file.

Skipped where the compiled reference is absent (it is prebuilt in the dev container and
travels to the GPU box with the snapshot).  CPU only.
"""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.skipif(not (ol.available("ref") and ol.available("port")),
                                reason="needs both oracle libraries")



@pytest.fixture(scope="module")
def libs():
    return ol.load("ref"), ol.load("port")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, what
    bad = np.flatnonzero(bits(a).ravel() != bits(b).ravel())
    assert bad.size == 0, (f"{what}: {bad.size} of {a.size} differ; first at {bad[0]}: "
                           f"{a.ravel()[bad[0]]!r} vs {b.ravel()[bad[0]]!r}")


def test_bsinc_tables_bit_exact(libs):
    ref, port = libs
    for which in (12, 24, 48):
        a, b = ref.bsinc_table(which), port.bsinc_table(which)
        assert a["m"] == b["m"] and a["filterOffset"] == b["filterOffset"]
        assert np.float32(a["scaleBase"]) == np.float32(b["scaleBase"])
        assert np.float32(a["scaleRange"]) == np.float32(b["scaleRange"])
        assert_bit_equal(a["tab"], b["tab"], f"bsinc{which} table")


def test_cubic_tables_bit_exact(libs):
    ref, port = libs
    for which in (0, 1):
        assert_bit_equal(ref.cubic_table(which), port.cubic_table(which), f"cubic table {which}")


@pytest.mark.parametrize("resampler", range(10))
def test_prepare_resampler(libs, resampler):
    ref, port = libs
    for inc in (1, 30000, 60211, 65536, 65537, 70000, 98304, 131072, 200000, 400000, 655360):
        a, b = ref.prepare_resampler(resampler, inc), port.prepare_resampler(resampler, inc)
        for f in ("kind", "table", "m", "l", "filter_offset"):
            assert getattr(a, f) == getattr(b, f), (resampler, inc, f)
        assert np.float32(a.sf).tobytes() == np.float32(b.sf).tobytes(), (resampler, inc)


@pytest.mark.parametrize("simd", [0, 1])
@pytest.mark.parametrize("resampler", range(10))
def test_resample_bit_exact(libs, resampler, simd):
    ref, port = libs
    rng = np.random.default_rng(100 + resampler)
    for inc, frac, n in ((60211, 0, 1024), (60211, 12345, 1021), (65536, 1, 7), (30000, 65535, 1024),
                         (90000, 777, 512), (131072, 40000, 600), (250000, 3, 301)):
        need = ((n * inc + frac) >> 16) + 64
        src = rng.uniform(-1, 1, need + 64).astype(np.float32)
        src[5] = 1e-41  # a denormal: FTZ/DAZ must agree
        for L in (ref, port):
            L.L.oal_set_simd(simd)
        assert_bit_equal(ref.resample(resampler, inc, src, frac, n),
                         port.resample(resampler, inc, src, frac, n), f"rs{resampler} inc{inc}")
    for L in (ref, port):
        L.L.oal_set_simd(1)


@pytest.mark.parametrize("simd", [0, 1])
def test_mix_bit_exact(libs, simd):
    ref, port = libs
    rng = np.random.default_rng(7)
    for L in (ref, port):
        L.L.oal_set_simd(simd)
    for nlines, n, counter, outpos in ((3, 1024, 0, 0), (5, 1024, 64, 0), (9, 1000, 64, 24),
                                       (4, 40, 64, 0), (2, 1000, 64, 3), (5, 61, 61, 0)):
        inp = rng.uniform(-1, 1, n).astype(np.float32)
        base = rng.uniform(-1, 1, (nlines, 1024)).astype(np.float32)
        cur0 = rng.uniform(0, 1, nlines).astype(np.float32)
        tgt = rng.uniform(0, 1, nlines).astype(np.float32)
        tgt[0] = 1e-6          # below GainSilenceThreshold
        if nlines > 2:
            tgt[2] = cur0[2]   # |step| <= epsilon
        outs = []
        for L in (ref, port):
            out, cur = base.copy(), cur0.copy()
            L.mix(inp, out, cur, tgt, counter, outpos)
            outs.append((out, cur))
        assert_bit_equal(outs[0][0], outs[1][0], "mix out")
        assert_bit_equal(outs[0][1], outs[1][1], "mix current gains")
    for L in (ref, port):
        L.L.oal_set_simd(1)


@pytest.mark.parametrize("simd", [0, 1])
def test_mix_hrtf_bit_exact(libs, simd):
    ref, port = libs
    rng = np.random.default_rng(11)
    for L in (ref, port):
        L.L.oal_set_simd(simd)
    for irsize, n, delay in ((64, 1024, (10, 13)), (32, 500, (0, 63)), (9, 64, (63, 0)),
                             (128, 1024, (5, 7))):
        inp = rng.uniform(-1, 1, n + 64).astype(np.float32)
        co = np.zeros((128, 2), np.float32)
        co[:irsize] = rng.uniform(-0.5, 0.5, (irsize, 2))
        oldco = np.zeros((128, 2), np.float32)
        oldco[:irsize] = rng.uniform(-0.5, 0.5, (irsize, 2))
        acc0 = rng.uniform(-1, 1, (1024 + 128, 2)).astype(np.float32)
        res = []
        for L in (ref, port):
            acc = acc0.copy()
            L.mix_hrtf(inp, acc, irsize, co, delay, 0.3, 0.001, n)
            fm = min(n, 64)
            L.mix_hrtf_blend(inp, acc, irsize, oldco, (3, 9), 0.25, co, delay, 0.3 / fm, fm)
            L.mix_hrtf_blend(inp, acc, irsize, oldco, (3, 9), 1e-6, co, delay, 1e-8, fm)
            res.append(acc)
        assert_bit_equal(res[0], res[1], f"hrtf irsize {irsize}")
    for L in (ref, port):
        L.L.oal_set_simd(1)


def test_biquad_bit_exact(libs):
    ref, port = libs
    rng = np.random.default_rng(3)
    import ctypes as C
    src_all = rng.uniform(-1, 1, 4096).astype(np.float32)
    states = []
    outs = []
    for L in (ref, port):
        lp, hp = ol.Biquad(), ol.Biquad()
        L.L.oal_biquad_reset(C.byref(lp))
        L.L.oal_biquad_reset(C.byref(hp))
        out = []
        st = []
        gains = [(0.5, 1.0), (0.5, 1.0), (0.25, 0.9), (0.25, 0.9), (0.9, 0.3), (0.9, 0.3)]
        lens = [1024, 37, 1024, 500, 1000, 3]
        pos = 0
        for (ghf, glf), n in zip(gains, lens):
            L.L.oal_biquad_set_params_from_slope(C.byref(lp), 0, 5000 / 48000, ghf, 1.0)
            L.L.oal_biquad_set_params_from_slope(C.byref(hp), 1, 250 / 48000, glf, 1.0)
            st.append(lp.as_tuple() + hp.as_tuple())
            src = src_all[pos:pos + n].copy()
            dst = np.zeros(n, np.float32)
            L.L.oal_biquad_dual_process(C.byref(lp), C.byref(hp), src.ctypes.data_as(ol.f32p),
                                        dst.ctypes.data_as(ol.f32p), n)
            out.append(dst)
            st.append(lp.as_tuple() + hp.as_tuple())
            pos += n
        L.L.oal_biquad_clear(C.byref(lp))
        st.append(lp.as_tuple())
        outs.append(np.concatenate(out))
        states.append(st)
    assert_bit_equal(outs[0], outs[1], "biquad out")
    for a, b in zip(states[0], states[1]):
        assert np.array(a[:12], np.float32).tobytes() == np.array(b[:12], np.float32).tobytes()
        assert a[12] == b[12]


def test_splitter_and_direct_hrtf_bit_exact(libs):
    ref, port = libs
    import ctypes as C
    rng = np.random.default_rng(5)
    nch, irsize = 4, 64
    inp = rng.uniform(-1, 1, (nch, 1024)).astype(np.float32)
    cc = np.zeros((nch, 128, 2), np.float32)
    cc[:, :irsize] = rng.uniform(-0.3, 0.3, (nch, irsize, 2))
    hf = np.array([1.0, 0.7, 0.7, 0.7], np.float32)
    res = []
    for L in (ref, port):
        sp = []
        for _ in range(nch):
            s = ol.Splitter()
            L.L.oal_splitter_init(C.byref(s), 400.0 / 48000.0)
            sp.append(s)
        left = rng.uniform(-1, 1, 1024).astype(np.float32) * 0 + 0.25
        right = left.copy()
        acc = np.zeros((1024 + 128, 2), np.float32)
        outs = []
        for n in (1024, 600, 1024):
            sp = L.mix_direct_hrtf(left, right, inp, acc, sp, hf, cc, irsize, n)
            outs += [left.copy(), right.copy(), acc.copy().ravel()]
        outs.append(np.array([[s.coeff, s.lp_z1, s.lp_z2, s.ap_z1] for s in sp], np.float32).ravel())
        res.append(np.concatenate(outs))
    assert_bit_equal(res[0], res[1], "MixDirectHrtf")


def test_hrtf_store_and_getcoeffs_bit_exact(libs, mhr_paths):
    for MHR in mhr_paths:
        _check_hrtf_store(libs, MHR)


def _check_hrtf_store(libs, MHR):
    ref, port = libs
    ia, ib = ref.hrtf_load(MHR), port.hrtf_load(MHR)
    for f in ("sample_rate", "ir_size", "num_fields", "num_elevs", "num_irs"):
        assert getattr(ia, f) == getattr(ib, f)
    ra, rb = ref.hrtf_raw(), port.hrtf_raw()
    for k in ("field_evcount", "elev_azcount", "elev_iroffset", "delays"):
        assert np.array_equal(ra[k], rb[k]), k
    assert_bit_equal(ra["field_distance"], rb["field_distance"])
    assert_bit_equal(ra["coeffs"], rb["coeffs"], "HRIR store")
    rng = np.random.default_rng(9)
    dirs = [(0.0, 0.0, 2.0, 0.0), (np.pi / 2, 0.0, 1.0, 0.0), (-np.pi / 2, 3.0, 0.1, 1.0),
            (0.3, -np.pi, 5.0, 6.2), (0.3, np.pi, 5.0, 0.0)]
    dirs += [(np.arcsin(rng.uniform(-1, 1)), rng.uniform(-np.pi, np.pi), rng.uniform(0.05, 5),
              rng.uniform(0, 2 * np.pi)) for _ in range(200)]
    for ev, az, dist, spread in dirs:
        ca, da = ref.hrtf_get_coeffs(ev, az, dist, spread)
        cb, db = port.hrtf_get_coeffs(ev, az, dist, spread)
        assert da == db, (ev, az)
        assert_bit_equal(ca, cb, f"getCoeffs {ev} {az}")


from scenes import SCENES, run_scene


@pytest.mark.parametrize("idx", range(len(SCENES)))
def test_scene_voice_mix_bit_exact(libs, mhr_paths, idx):
    ref, port = libs
    cfg = dict(SCENES[idx])
    MHR = mhr_paths[-1]
    fa, ia = run_scene(ref, MHR, rng_seed=idx + 1, **cfg)
    fb, ib = run_scene(port, MHR, rng_seed=idx + 1, **cfg)
    assert ia == ib, "integer voice state (positions, play state, delays, counters)"
    assert_bit_equal(fa, fb, f"scene {idx}")


@pytest.mark.parametrize("ir_size", [24, 32, 128])
def test_other_hrir_lengths_bit_exact(libs, tmp_path, ir_size):
    """Data sets whose IrSize is not 64 (what tests/test_gpu_parity.py::test_hrir_lengths_other_than_64
    checks the GPU against): store, getCoeffs and a moving-source HRTF scene, restatement against the
    compiled reference."""
    from oalgpu import synth
    ref, port = libs
    path = synth.write_synth_mhr(str(tmp_path / f"ir{ir_size}.mhr"), ir_size=ir_size)
    _check_hrtf_store(libs, path)
    cfg = dict(hrtf=True, fmt=ol.FMT_FLOAT, resampler=ol.RS_BSINC24, steps=[60211, 70000], n_updates=4, nvoices=14)
    fa, ia = run_scene(ref, path, rng_seed=7, **cfg)
    fb, ib = run_scene(port, path, rng_seed=7, **cfg)
    assert ia == ib
    assert_bit_equal(fa, fb, f"IrSize {ir_size} scene")


MULTI_FIELD = [(1400, [1, 12, 24, 36, 24, 12, 1]), (900, [1, 8, 16, 24, 30, 24, 16, 8, 1]), (300, [1, 6, 12, 6, 1])]


@pytest.mark.parametrize("stereo", [False, True], ids=["left_only", "left_right"])
def test_multi_field_data_sets_bit_exact(libs, tmp_path, stereo):
    """A data set with three field depths of different elevation / azimuth layouts (HrtfStore::getCoeffs picks the field
    by distance, core/hrtf.cpp:198-207) and one that stores both ears (no mirroring): store, getCoeffs and a scene
    whose sources sit at distances on both sides of every field boundary."""
    from oalgpu import synth
    ref, port = libs
    path = synth.write_synth_mhr(str(tmp_path / "fields.mhr"), fields=MULTI_FIELD, stereo=stereo, ir_size=32)
    info = ref.hrtf_load(path)
    assert info.num_fields == 3 and info.num_elevs == 7 + 9 + 5
    _check_hrtf_store(libs, path)
    for dist in (0.1, 0.3, 0.31, 0.9, 0.95, 1.4, 2.0):
        ca, da = ref.hrtf_get_coeffs(0.2, 1.0, dist, 0.0)
        cb, db = port.hrtf_get_coeffs(0.2, 1.0, dist, 0.0)
        assert da == db
        assert_bit_equal(ca, cb, f"getCoeffs at {dist} m")
    cfg = dict(hrtf=True, fmt=ol.FMT_FLOAT, resampler=ol.RS_BSINC24, steps=[60211], n_updates=3, nvoices=10,
               distances=[0.1, 0.3, 0.5, 0.9, 1.0, 1.4, 2.0])
    fa, ia = run_scene(ref, path, rng_seed=11, **cfg)
    fb, ib = run_scene(port, path, rng_seed=11, **cfg)
    assert ia == ib
    assert_bit_equal(fa, fb, "multi-field scene")
