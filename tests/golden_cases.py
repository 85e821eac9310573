"""Deterministic known-answer cases, run against any oracle_lib-compatible library.

``collect(L, mhr)`` returns {name: float32 array | int list}.  tests/golden/make_golden.py runs
it on the COMPILED REFERENCE (oracle/_ref) and stores sha256 digests (+ small arrays);
tests/test_oracle_golden.py re-runs it on the C restatement and compares bit-exact.
Inputs are regenerated from seeds, never stored.
"""
import ctypes as C
import hashlib

import numpy as np

import oracle_lib as ol
from scenes import SCENES, run_scene


def digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()


def collect(L, mhr):
    out = {}
    L.L.oal_set_simd(1)
    for which in (12, 24, 48):
        t = L.bsinc_table(which)
        out[f"bsinc{which}.tab"] = t["tab"]
        out[f"bsinc{which}.hdr"] = np.array(t["m"] + t["filterOffset"], np.int64)
        out[f"bsinc{which}.scale"] = np.array([t["scaleBase"], t["scaleRange"]], np.float32)
    out["cubic.spline"] = L.cubic_table(0)
    out["cubic.gaussian"] = L.cubic_table(1)
    rng = np.random.default_rng(2024)
    src = rng.uniform(-1, 1, 12000).astype(np.float32)
    for rs in range(10):
        for inc, frac, n in ((60211, 4660, 1024), (150000, 9, 333)):
            for simd in (0, 1):
                L.L.oal_set_simd(simd)
                out[f"resample.{rs}.{inc}.simd{simd}"] = L.resample(rs, inc, src, frac, n)
    L.L.oal_set_simd(1)
    # Mix
    inp = rng.uniform(-1, 1, 1024).astype(np.float32)
    lines = np.zeros((5, 1024), np.float32)
    cur = np.array([0.1, 0.2, 0.0, 0.5, 0.25], np.float32)
    tgt = np.array([0.3, 0.2, 1e-6, 0.0, 0.75], np.float32)
    L.mix(inp, lines, cur, tgt, 64, 0)
    L.mix(inp[:500], lines, cur, tgt * 0.5, 64, 100)
    out["mix.lines"] = lines
    out["mix.cur"] = cur
    # HRTF mixers
    co = np.zeros((128, 2), np.float32)
    co[:64] = rng.uniform(-0.5, 0.5, (64, 2))
    oldco = np.zeros((128, 2), np.float32)
    oldco[:64] = rng.uniform(-0.5, 0.5, (64, 2))
    hin = rng.uniform(-1, 1, 1024 + 64).astype(np.float32)
    acc = np.zeros((1024 + 128, 2), np.float32)
    L.mix_hrtf_blend(hin, acc, 64, oldco, (7, 30), 0.4, co, (12, 3), 0.5 / 64, 64)
    L.mix_hrtf(hin[64:], acc[64:], 64, co, (12, 3), 0.5, 0.0001, 960)
    out["hrtf.accum"] = acc
    # biquad
    lp, hp = ol.Biquad(), ol.Biquad()
    L.L.oal_biquad_reset(C.byref(lp))
    L.L.oal_biquad_reset(C.byref(hp))
    outs = []
    for ghf, n in ((0.5, 1024), (0.1, 1024), (0.1, 300)):
        L.L.oal_biquad_set_params_from_slope(C.byref(lp), 0, 5000 / 48000, ghf, 1.0)
        L.L.oal_biquad_set_params_from_slope(C.byref(hp), 1, 250 / 48000, 0.8, 1.0)
        dst = np.zeros(n, np.float32)
        L.L.oal_biquad_dual_process(C.byref(lp), C.byref(hp), inp[:n].ctypes.data_as(ol.f32p),
                                    dst.ctypes.data_as(ol.f32p), n)
        outs.append(dst)
    out["biquad.out"] = np.concatenate(outs)
    out["biquad.state"] = np.array(lp.as_tuple()[:12] + hp.as_tuple()[:12], np.float32)
    out["biquad.counter"] = np.array([lp.counter, hp.counter], np.int64)
    # direct HRTF
    din = rng.uniform(-1, 1, (4, 1024)).astype(np.float32)
    cc = np.zeros((4, 128, 2), np.float32)
    cc[:, :64] = rng.uniform(-0.3, 0.3, (4, 64, 2))
    sp = []
    for _ in range(4):
        s = ol.Splitter()
        L.L.oal_splitter_init(C.byref(s), 400.0 / 48000.0)
        sp.append(s)
    left = np.zeros(1024, np.float32)
    right = np.zeros(1024, np.float32)
    dacc = np.zeros((1024 + 128, 2), np.float32)
    sp = L.mix_direct_hrtf(left, right, din, dacc, sp, [1.0, 0.8, 0.8, 0.8], cc, 64, 1024)
    sp = L.mix_direct_hrtf(left, right, din, dacc, sp, [1.0, 0.8, 0.8, 0.8], cc, 64, 512)
    out["direct.left"] = left
    out["direct.right"] = right
    out["direct.accum"] = dacc
    # HRTF data set + getCoeffs (synthetic .mhr in the reference's v3 format)
    L.hrtf_load(mhr)
    raw = L.hrtf_raw()
    out["mhr.coeffs"] = raw["coeffs"]
    out["mhr.delays"] = raw["delays"].astype(np.int64)
    gc = []
    gd = []
    for k in range(64):
        ev = np.float32(np.arcsin(rng.uniform(-1, 1)))
        az = np.float32(rng.uniform(-np.pi, np.pi))
        c_, d_ = L.hrtf_get_coeffs(float(ev), float(az), float(np.float32(0.5 + k * 0.05)),
                                   float(np.float32((k % 5) * 1.1)))
        gc.append(c_)
        gd.append(d_)
    out["getcoeffs.coeffs"] = np.stack(gc)
    out["getcoeffs.delays"] = np.array(gd, np.int64)
    # full Voice::mix scenes
    for idx, cfg in enumerate(SCENES):
        f, ints = run_scene(L, mhr, rng_seed=idx + 1, **cfg)
        out[f"scene{idx}.float"] = f
        out[f"scene{idx}.int"] = np.array(ints, np.int64)
    return out
