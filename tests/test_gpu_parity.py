"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same inputs.

Oracle = the compiled reference (oracle/_ref/liboalref.so) when it travelled with the snapshot,
else the C restatement (oracle/liboalport.so) -- the two are pinned bit-equal to each other by
test_oracle_pin.py / test_oracle_golden.py.

Tolerances (stated once, used everywhere below):
  * EXACT mode, single voice / single kernel call: BIT-EXACT (floats compared as uint32).
  * integer state (positions, fractions, play state, delays, filter counters): always exact.
  * sums over several voices (any mode): the GPU adds voices in a different order than the
    serial CPU loop, so |gpu-ref| <= MULTI_RTOL * max|ref| + MULTI_ATOL per bus.
  * FAST mode: FMA in the FIR loops; same bound as the multi-voice one.
"""
import ctypes as C
import os

import numpy as np
import pytest

import golden_cases
import oracle_lib as ol
import oalgpu
from scenes import SCENES, run_scene

pytestmark = pytest.mark.gpu

MULTI_RTOL = 2e-5
MULTI_ATOL = 1e-7


@pytest.fixture(scope="module")
def oracle():
    which = "ref" if ol.available("ref") else "port"
    if not ol.available(which):
        pytest.skip("no oracle library built")
    L = ol.load(which)
    L.L.oal_set_simd(1)
    return L


@pytest.fixture(scope="module")
def exact():
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    return oalgpu.Api(oalgpu.MATH_EXACT)


@pytest.fixture(scope="module")
def fast():
    return oalgpu.Api(oalgpu.MATH_FAST)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, what
    bad = np.flatnonzero(bits(a).ravel() != bits(b).ravel())
    assert bad.size == 0, (f"{what}: {bad.size}/{a.size} differ; first at {bad[0]}: "
                           f"{a.ravel()[bad[0]]!r} vs {b.ravel()[bad[0]]!r}")


def assert_close(gpu, ref, what="", rtol=MULTI_RTOL, atol=MULTI_ATOL):
    gpu = np.asarray(gpu, np.float64)
    ref = np.asarray(ref, np.float64)
    bound = rtol * np.max(np.abs(ref)) + atol
    err = np.max(np.abs(gpu - ref)) if gpu.size else 0.0
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e} (max|ref| {np.max(np.abs(ref)):.3e})"


# ------------------------------------------------------------------ per-call kernels, EXACT
@pytest.mark.parametrize("resampler", range(10))
def test_resample_bit_exact(oracle, exact, resampler):
    rng = np.random.default_rng(100 + resampler)
    for inc, frac, n in ((60211, 0, 1024), (60211, 12345, 1021), (65536, 1, 7), (30000, 65535, 1024),
                         (90000, 777, 512), (131072, 40000, 600), (250000, 3, 301), (655360, 9, 100)):
        need = ((n * inc + frac) >> 16) + 64
        src = rng.uniform(-1, 1, need + 64).astype(np.float32)
        src[5] = 1e-41      # denormal input: both sides flush
        assert_bit_equal(exact.resample(resampler, inc, src, frac, n),
                         oracle.resample(resampler, inc, src, frac, n), f"rs{resampler} inc{inc} n{n}")


def test_resample_fast_within_tolerance(oracle, fast):
    rng = np.random.default_rng(5)
    for rs in (ol.RS_SPLINE, ol.RS_FAST_BSINC24, ol.RS_BSINC24, ol.RS_BSINC48):
        for inc in (60211, 150000):
            src = rng.uniform(-1, 1, 4096).astype(np.float32)
            assert_close(fast.resample(rs, inc, src, 77, 1024), oracle.resample(rs, inc, src, 77, 1024),
                         f"fast rs{rs}")


def test_mix_bit_exact(oracle, exact):
    rng = np.random.default_rng(7)
    for nlines, n, counter, outpos in ((3, 1024, 0, 0), (5, 1024, 64, 0), (9, 1000, 64, 24), (4, 40, 64, 0),
                                       (2, 1000, 64, 3), (5, 61, 61, 0), (32, 512, 64, 512)):
        inp = rng.uniform(-1, 1, n).astype(np.float32)
        base = rng.uniform(-1, 1, (nlines, 1024)).astype(np.float32)
        cur0 = rng.uniform(0, 1, nlines).astype(np.float32)
        tgt = rng.uniform(0, 1, nlines).astype(np.float32)
        tgt[0] = 1e-6
        if nlines > 2:
            tgt[2] = cur0[2]
        res = []
        for L in (exact, oracle):
            out, cur = base.copy(), cur0.copy()
            L.mix(inp, out, cur, tgt, counter, outpos)
            res.append((out, cur))
        assert_bit_equal(res[0][0], res[1][0], "mix out")
        assert_bit_equal(res[0][1], res[1][1], "mix current gains")


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_mix_hrtf_and_blend(oracle, exact, fast, mode):
    api = exact if mode == "exact" else fast
    rng = np.random.default_rng(11)
    for irsize, n, delay in ((64, 1024, (10, 13)), (32, 500, (0, 63)), (9, 64, (63, 0)), (128, 1024, (5, 7))):
        inp = rng.uniform(-1, 1, n + 64).astype(np.float32)
        co = np.zeros((128, 2), np.float32)
        co[:irsize] = rng.uniform(-0.5, 0.5, (irsize, 2))
        oldco = np.zeros((128, 2), np.float32)
        oldco[:irsize] = rng.uniform(-0.5, 0.5, (irsize, 2))
        acc0 = rng.uniform(-1, 1, (1024 + 128, 2)).astype(np.float32)
        res = []
        for L in (api, oracle):
            acc = acc0.copy()
            L.mix_hrtf(inp, acc, irsize, co, delay, 0.3, 0.001, n)
            fm = min(n, 64)
            L.mix_hrtf_blend(inp, acc, irsize, oldco, (3, 9), 0.25, co, delay, 0.3 / fm, fm)
            L.mix_hrtf_blend(inp, acc, irsize, oldco, (3, 9), 1e-6, co, delay, 1e-8, fm)
            res.append(acc)
        if mode == "exact":
            assert_bit_equal(res[0], res[1], f"hrtf irsize {irsize}")
        else:
            assert_close(res[0], res[1], f"hrtf fast irsize {irsize}")


def test_biquad_bit_exact(oracle, exact):
    rng = np.random.default_rng(3)
    src_all = rng.uniform(-1, 1, 4096).astype(np.float32)
    a_lp, a_hp, b_lp, b_hp = oalgpu.Biquad(), oalgpu.Biquad(), ol.Biquad(), ol.Biquad()
    for f in (a_lp, a_hp):
        oalgpu.lib.oalgpu_biquad_reset(C.byref(f))
    for f in (b_lp, b_hp):
        oracle.L.oal_biquad_reset(C.byref(f))
    pos = 0
    for (ghf, glf), n in zip([(0.5, 1.0), (0.5, 1.0), (0.25, 0.9), (0.25, 0.9), (0.9, 0.3), (0.9, 0.3)],
                             [1024, 37, 1024, 500, 1000, 3]):
        oalgpu.lib.oalgpu_biquad_set_params_from_slope(C.byref(a_lp), 0, 5000 / 48000, ghf, 1.0)
        oalgpu.lib.oalgpu_biquad_set_params_from_slope(C.byref(a_hp), 1, 250 / 48000, glf, 1.0)
        oracle.L.oal_biquad_set_params_from_slope(C.byref(b_lp), 0, 5000 / 48000, ghf, 1.0)
        oracle.L.oal_biquad_set_params_from_slope(C.byref(b_hp), 1, 250 / 48000, glf, 1.0)
        src = src_all[pos:pos + n].copy()
        got = exact.biquad_dual_process(a_lp, a_hp, src)
        want = np.zeros(n, np.float32)
        oracle.L.oal_biquad_dual_process(C.byref(b_lp), C.byref(b_hp), src.ctypes.data_as(ol.f32p),
                                         want.ctypes.data_as(ol.f32p), n)
        assert_bit_equal(got, want, "biquad out")
        for x, y in ((a_lp, b_lp), (a_hp, b_hp)):
            assert np.array(x.as_tuple()[:12], np.float32).tobytes() == np.array(y.as_tuple()[:12], np.float32).tobytes()
            assert x.counter == y.counter
        pos += n


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_mix_direct_hrtf(oracle, exact, fast, mode):
    api = exact if mode == "exact" else fast
    rng = np.random.default_rng(5)
    nch, irsize = 4, 64
    inp = rng.uniform(-1, 1, (nch, 1024)).astype(np.float32)
    cc = np.zeros((nch, 128, 2), np.float32)
    cc[:, :irsize] = rng.uniform(-0.3, 0.3, (nch, irsize, 2))
    hf = np.array([1.0, 0.7, 0.7, 0.7], np.float32)
    res = []
    for L in (api, oracle):
        sp = []
        for _ in range(nch):
            s = ol.Splitter()
            oracle.L.oal_splitter_init(C.byref(s), 400.0 / 48000.0)
            sp.append(s)
        left = np.full(1024, 0.25, np.float32)
        right = left.copy()
        acc = np.zeros((1024 + 128, 2), np.float32)
        outs = []
        for n in (1024, 600, 1024):
            sp = L.mix_direct_hrtf(left, right, inp, acc, sp, hf, cc, irsize, n)
            outs += [left.copy(), right.copy(), acc.copy().ravel()]
        outs.append(np.array([[s.coeff, s.lp_z1, s.lp_z2, s.ap_z1] for s in sp], np.float32).ravel())
        res.append(np.concatenate(outs))
    if mode == "exact":
        assert_bit_equal(res[0], res[1], "MixDirectHrtf")
    else:
        assert_close(res[0], res[1], "MixDirectHrtf fast")


def test_get_coeffs_on_gpu_bit_exact(oracle, exact, synth_mhr):
    oracle.hrtf_load(synth_mhr)
    exact.hrtf_load(synth_mhr)
    sc = exact.make_scene(num_dry=4, num_real=2, hrtf=True)
    raw_a, raw_b = sc.hrtf_raw(), oracle.hrtf_raw()
    for k in ("field_evcount", "elev_azcount", "elev_iroffset", "delays"):
        assert np.array_equal(raw_a[k], raw_b[k]), k
    assert_bit_equal(raw_a["coeffs"], raw_b["coeffs"], "parsed HRIR store")
    rng = np.random.default_rng(9)
    dirs = [(0.0, 0.0, 2.0, 0.0), (np.pi / 2, 0.0, 1.0, 0.0), (-np.pi / 2, 3.0, 0.1, 1.0),
            (0.3, -np.pi, 5.0, 6.2), (0.3, np.pi, 5.0, 0.0)]
    dirs += [(np.arcsin(rng.uniform(-1, 1)), rng.uniform(-np.pi, np.pi), rng.uniform(0.05, 5),
              rng.uniform(0, 2 * np.pi)) for _ in range(300)]
    dirs = np.array(dirs, np.float32)
    co, de = sc.hrtf_get_coeffs(dirs)
    for i, d in enumerate(dirs):
        c_ref, d_ref = oracle.hrtf_get_coeffs(float(d[0]), float(d[1]), float(d[2]), float(d[3]))
        assert tuple(de[i]) == d_ref, (i, d)
        assert_bit_equal(co[i], c_ref, f"getCoeffs {i}")
    sc.close()


@pytest.mark.parametrize("stereo", [False, True], ids=["left_only", "left_right"])
def test_get_coeffs_multi_field_sets(oracle, exact, fast, tmp_path, stereo):
    """three field depths with different layouts, and a set that stores both ears: the parsed store, and getCoeffs on
    the GPU (the parameter kernel's LDS copy of the index tables) for directions at distances around every field
    boundary, bit for bit"""
    from oalgpu import synth
    fields = [(1400, [1, 12, 24, 36, 24, 12, 1]), (900, [1, 8, 16, 24, 30, 24, 16, 8, 1]), (300, [1, 6, 12, 6, 1])]
    path = synth.write_synth_mhr(str(tmp_path / "fields.mhr"), fields=fields, stereo=stereo, ir_size=32)
    oracle.hrtf_load(path)
    exact.hrtf_load(path)
    sc = exact.make_scene(num_dry=4, num_real=2, hrtf=True)
    raw_a, raw_b = sc.hrtf_raw(), oracle.hrtf_raw()
    for k in ("field_evcount", "elev_azcount", "elev_iroffset", "delays"):
        assert np.array_equal(raw_a[k], raw_b[k]), k
    assert_bit_equal(raw_a["coeffs"], raw_b["coeffs"], "parsed HRIR store")
    rng = np.random.default_rng(19)
    dirs = [(0.2, 1.0, d, 0.0) for d in (0.05, 0.3, 0.3000001, 0.31, 0.9, 0.91, 1.4, 1.41, 3.0)]
    dirs += [(np.arcsin(rng.uniform(-1, 1)), rng.uniform(-np.pi, np.pi), rng.uniform(0.05, 2.0), rng.uniform(0, 2 * np.pi))
             for _ in range(200)]
    dirs = np.array(dirs, np.float32)
    co, de = sc.hrtf_get_coeffs(dirs)
    fields_seen = set()
    for i, d in enumerate(dirs):
        c_ref, d_ref = oracle.hrtf_get_coeffs(float(d[0]), float(d[1]), float(d[2]), float(d[3]))
        assert tuple(de[i]) == d_ref, (i, d)
        assert_bit_equal(co[i], c_ref, f"getCoeffs {i}")
        fields_seen.add(0 if d[2] >= 1.4 else 1 if d[2] >= 0.9 else 2)
    assert fields_seen == {0, 1, 2}
    sc.close()
    # the voice path: moving sources whose parameter records carry those distances (ApplyParamsKernel)
    cfg = dict(hrtf=True, fmt=ol.FMT_FLOAT, resampler=ol.RS_BSINC24, steps=[60211], n_updates=3, nvoices=10,
               distances=[0.1, 0.3, 0.5, 0.9, 1.0, 1.4, 2.0])
    _cmp_scene(exact, oracle, path, cfg, 11, single=False)
    _cmp_scene(fast, oracle, path, cfg, 12, single=False)


# ------------------------------------------------------------------ golden vectors (EXACT)
def test_per_call_kernels_match_reference_golden(exact, synth_mhr):
    """Same cases as tests/golden_cases.py, per-call kernels only, against the committed
    vectors generated from the compiled reference."""
    small = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_small.npz"))
    rng = np.random.default_rng(2024)
    src = rng.uniform(-1, 1, 12000).astype(np.float32)
    for rs in range(10):
        for inc, frac, n in ((60211, 4660, 1024), (150000, 9, 333)):
            assert_bit_equal(exact.resample(rs, inc, src, frac, n), small[f"resample.{rs}.{inc}.simd1"],
                             f"golden resample {rs} {inc}")
    inp = rng.uniform(-1, 1, 1024).astype(np.float32)
    lines = np.zeros((5, 1024), np.float32)
    cur = np.array([0.1, 0.2, 0.0, 0.5, 0.25], np.float32)
    tgt = np.array([0.3, 0.2, 1e-6, 0.0, 0.75], np.float32)
    exact.mix(inp, lines, cur, tgt, 64, 0)
    exact.mix(inp[:500], lines, cur, tgt * 0.5, 64, 100)
    assert_bit_equal(cur, small["mix.cur"], "golden mix cur")
    import json
    manifest = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))["items"]
    assert golden_cases.digest(lines) == manifest["mix.lines"]["sha256"], "golden mix lines"
    co = np.zeros((128, 2), np.float32)
    co[:64] = rng.uniform(-0.5, 0.5, (64, 2))
    oldco = np.zeros((128, 2), np.float32)
    oldco[:64] = rng.uniform(-0.5, 0.5, (64, 2))
    hin = rng.uniform(-1, 1, 1024 + 64).astype(np.float32)
    acc = np.zeros((1024 + 128, 2), np.float32)
    exact.mix_hrtf_blend(hin, acc, 64, oldco, (7, 30), 0.4, co, (12, 3), 0.5 / 64, 64)
    # the second call of the golden case works on an offset view (AccumSamples.subspan(64))
    tail = np.zeros((1024 + 128, 2), np.float32)
    tail[:1024 + 64] = acc[64:]
    exact.mix_hrtf(hin[64:], tail, 64, co, (12, 3), 0.5, 0.0001, 960)
    acc[64:] = tail[:1024 + 64]
    assert_bit_equal(acc, small["hrtf.accum"], "golden hrtf accum")


# ------------------------------------------------------------------ scene level (Voice::mix)
def _cmp_scene(api, oracle, mhr, cfg, seed, single):
    fa, ia = run_scene(api, mhr, rng_seed=seed, **cfg)
    fb, ib = run_scene(oracle, mhr, rng_seed=seed, **cfg)
    assert ia == ib, "integer voice state (positions, play state, delays, counters) must be exact"
    if single:
        assert_bit_equal(fa, fb, "single-voice scene")
    else:
        assert_close(fa, fb, "multi-voice scene")


@pytest.mark.parametrize("idx", range(len(SCENES)))
def test_scene_single_voice_bit_exact(oracle, exact, synth_mhr, idx):
    cfg = dict(SCENES[idx])
    cfg["nvoices"] = 1
    for seed in (idx + 1, idx + 101):
        _cmp_scene(exact, oracle, synth_mhr, cfg, seed, single=True)


@pytest.mark.parametrize("idx", range(len(SCENES)))
def test_scene_multi_voice_exact_mode(oracle, exact, synth_mhr, idx):
    _cmp_scene(exact, oracle, synth_mhr, dict(SCENES[idx]), idx + 1, single=False)


@pytest.mark.parametrize("idx", range(len(SCENES)))
def test_scene_multi_voice_fast_mode(oracle, fast, synth_mhr, idx):
    _cmp_scene(fast, oracle, synth_mhr, dict(SCENES[idx]), idx + 1, single=False)


@pytest.mark.gpu
@pytest.mark.parametrize("ir_size", [24, 32, 128])
def test_hrir_lengths_other_than_64(oracle, fast, exact, tmp_path, ir_size):
    """Data sets whose IrSize is not 64: the 16-tap-segment form of the FIR (IrSize 24 and 32 are
    stored as 32 taps per voice filter) and the 128-tap variant of the wavefront kernel, with moving
    sources (old-filter pass, coefficient hand-over) -- FAST against the oracle, and one voice in
    EXACT mode bit for bit."""
    from oalgpu import synth
    path = synth.write_synth_mhr(str(tmp_path / f"ir{ir_size}.mhr"), ir_size=ir_size)
    cfg = dict(hrtf=True, fmt=ol.FMT_FLOAT, resampler=ol.RS_BSINC24, steps=[60211, 70000], n_updates=4, nvoices=14)
    _cmp_scene(fast, oracle, path, cfg, 7, single=False)
    _cmp_scene(exact, oracle, path, dict(cfg, nvoices=1), 7, single=True)


# ---- the staged parameter blocks + the pipelined update (what bench.py drives) -------------------
@pytest.mark.gpu
@pytest.mark.parametrize("nvoices", [64, 9])
def test_param_blocks_through_the_pipeline(oracle, fast, synth_mhr, nvoices):
    """oalgpu_param_block_create/apply + oalgpu_mix_update on an HRTF context (the two-stream
    update): voices all over the grid move every update (records in shuffled order), one more voice is
    updated through the immediate call in between, and every update's buses and the final voice
    states must match the oracle driven with the same parameters."""
    oracle.hrtf_load(synth_mhr)
    fast.hrtf_load(synth_mhr)
    rng = np.random.default_rng(41)
    cc = np.zeros((4, 128, 2), np.float32)
    cc[:, :64] = rng.uniform(-0.2, 0.2, (4, 64, 2))
    data = rng.uniform(-1, 1, 8000).astype(np.float32)

    def params(v, k):
        r = np.random.default_rng(1000 * v + k)
        return ol.make_voice_params(60211 if v % 3 else 52000, ol.RS_BSINC24,
                                    hrtf=(np.arcsin(r.uniform(-1, 1)), r.uniform(-np.pi, np.pi), 2.0, 0.0,
                                          10 ** (r.uniform(-50, -20) / 20)),
                                    direct_filter=ol.default_filter(active=v % 4 == 1, gain_hf=0.5))

    def build(lib, **kw):
        sc = lib.make_scene(num_dry=4, num_real=2, hrtf=True, **kw)
        sc.set_direct_hrtf(cc, [1.0, 0.8, 0.8, 0.8], 400.0 / 48000.0, 64)
        b = sc.add_buffer(data, ol.FMT_FLOAT, loop_start=0, loop_end=8000)
        for v in range(nvoices):
            sc.add_voice(b, looping=True, position=(v * 911) % 7000, frac=(v * 977) % 65536)
            sc.set_params(v, params(v, 0))
        return sc

    gsc = build(fast, max_voices=nvoices)
    osc = build(oracle)
    moving = [v for v in range(nvoices) if v % 3 != 1]
    order = np.random.default_rng(5).permutation(len(moving))
    updates = 6
    blocks = []
    for k in range(updates):
        vs = [moving[i] for i in order]
        arr = (oalgpu.VoiceParams * len(vs))()
        for i, v in enumerate(vs):
            src = params(v, k + 1)
            C_memmove(arr, i, src)
        blocks.append(gsc.param_block(vs, arr))
    got, want = [], []
    for k in range(updates):
        gsc.apply_block(blocks[k])
        if k == 3:      # an immediate update of a voice the block also touched: the later call wins
            gsc.set_params(moving[0], params(moving[0], 77))
        gsc.mix(1024, post_process=True)
        for v in moving:
            osc.set_params(v, params(v, k + 1))
        if k == 3:
            osc.set_params(moving[0], params(moving[0], 77))
        osc.mix(1024, post_process=True)
        if k in (0, 2, 3, 5):       # reading back drains the pipeline: not after every update
            got.append(np.concatenate([gsc.dry().ravel(), gsc.hrtf_accum().ravel()]))
        want.append(np.concatenate([osc.dry().ravel(), osc.hrtf_accum().ravel()]))
    want = [want[k] for k in (0, 2, 3, 5)]
    for a, b in zip(got, want):
        assert_close(a, b, "pipelined update")
    for v in range(nvoices):
        sa, sb = gsc.voice_state(v), osc.voice_state(v)
        assert (sa.play_state, sa.position, sa.position_frac, sa.hrtf_old_delay[0], sa.hrtf_old_delay[1]) == \
            (sb.play_state, sb.position, sb.position_frac, sb.hrtf_old_delay[0], sb.hrtf_old_delay[1]), v
    gsc.close(); osc.close()


def C_memmove(arr, i, src):
    import ctypes as C
    C.memmove(C.byref(arr, i * C.sizeof(src)), C.byref(src), C.sizeof(src))
