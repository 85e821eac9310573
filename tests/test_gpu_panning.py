"""Panning on the GPU: oalgpu_voice_set_pan = CalcDirectionCoeffs + ComputePanGains
(core/mixer.h:68-73, core/ambidefs.h:219-271, core/mixer.cpp:16-102) for the dry bus and the sends' slots.

Two scenes mix the same voices: one gets its line gains from the host -- AmbiMap.Scale * coeffs[AmbiMap.Index] *
gain with coeffs from the REFERENCE's CalcDirectionCoeffs (oracle), in float32 like ComputePanGains -- through
oalgpu_voice_set_params; the other gets directions, spreads and gains through oalgpu_voice_set_pan.  Without
spread every bus must agree bit for bit over several updates (the GPU evaluates the same polynomials in the same
order); with spread (cos / sqrt on the GPU against libm) to 1e-6 of the bus maximum.  Orders 1-4, scaled and
permuted AmbiMaps, sends with and without slots, EXACT and FAST contexts, HRTF contexts (sends only)."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
ACN_COUNT = {1: 4, 2: 9, 3: 16, 4: 25}


def run(mode, order, spread_on, hrtf, synth_mhr, use_pan):
    import oalgpu
    which = "ref" if ol.available("ref") else "port"
    L = ol.load(which)
    api = oalgpu.Api(oalgpu.MATH_EXACT if mode == "exact" else oalgpu.MATH_FAST)
    if hrtf:
        api.hrtf_load(synth_mhr)
    rng = np.random.default_rng(order * 10 + spread_on)
    ndry = 4 if hrtf else min(ACN_COUNT[order], 24)      # 24 + 2 x 4 wet lines = the 32 mix lines a context holds
    nvoices, sends = 9, 2
    sc = api.make_scene(num_dry=ndry, num_real=2 if hrtf else 0, num_sends=sends, num_slots=2, wet_channels=4, hrtf=hrtf,
                        max_voices=nvoices)
    if hrtf:
        cc = np.zeros((4, 128, 2), np.float32)
        cc[:, :64] = rng.uniform(-0.2, 0.2, (4, 64, 2))
        sc.set_direct_hrtf(cc, [1.0, 0.8, 0.8, 0.8], 400.0 / 48000.0, 64)
    # a permuted, scaled dry map (FuMa-like devices reorder and rescale) and two different slot maps
    dry_idx = rng.permutation(ACN_COUNT[order])[:ndry].astype(np.uint8) if not hrtf else np.arange(4, dtype=np.uint8)
    dry_scale = rng.uniform(0.5, 1.5, ndry).astype(np.float32)
    wet_idx = [np.arange(4, dtype=np.uint8), np.array([0, 3, 1, 2], np.uint8)]
    wet_scale = [np.ones(4, np.float32), rng.uniform(0.5, 1.5, 4).astype(np.float32)]
    if use_pan:
        sc.set_ambi_map(dry_idx, dry_scale)
        for s in range(2):
            sc.set_slot_ambi_map(s, wet_idx[s], wet_scale[s])
    buf = sc.add_buffer(rng.uniform(-1, 1, 6000).astype(np.float32), ol.FMT_FLOAT, loop_start=0, loop_end=6000)
    for v in range(nvoices):
        sc.add_voice(buf, looping=True, position=(v * 577) % 5000, frac=(v * 4001) % 65536)

    def scene_params(k):
        r = np.random.default_rng(1000 * k + 7)
        out = []
        for v in range(nvoices):
            d = r.standard_normal(3)
            d = (d / np.linalg.norm(d)).astype(np.float32)
            spread = np.float32(r.uniform(0.1, 6.0)) if (spread_on and v % 2 == 0) else np.float32(0.0)
            dry_gain = np.float32(r.uniform(0.05, 0.6))
            send_gain = r.uniform(0.05, 0.5, 6).astype(np.float32)
            slots = [(v + i) % 3 - 1 for i in range(sends)]           # -1: no slot
            out.append((d, spread, dry_gain, send_gain, slots))
        return out

    outs = []
    for k in range(4):
        if k in (0, 2):
            voices, pans = [], []
            for v, (d, spread, dry_gain, send_gain, slots) in enumerate(scene_params(k)):
                coeffs = L.direction_coeffs(d, float(spread))
                dry = (dry_scale * coeffs[dry_idx]) * dry_gain                                      # ComputePanGains
                snd = []
                for i in range(sends):
                    g = np.zeros(4, np.float32)
                    if slots[i] >= 0:
                        g = (wet_scale[slots[i]] * coeffs[wet_idx[slots[i]]]) * send_gain[i]
                    snd.append((slots[i], g if not use_pan else np.zeros(4, np.float32), ol.default_filter(active=(v + i) % 2, gain_hf=0.6)))
                kw = dict(direct_filter=ol.default_filter(active=v % 2, gain_hf=0.5), sends=snd)
                if hrtf:
                    p = ol.make_voice_params(60211, ol.RS_BSINC24, hrtf=(0.3, 0.5 * v, 2.0, 0.0, 0.2), **kw)
                else:
                    p = ol.make_voice_params(60211, ol.RS_BSINC24, dry_gains=(np.zeros(ndry) if use_pan else dry), **kw)
                sc.set_params(v, p)
                voices.append(v)
                pans.append(list(d) + [spread, dry_gain] + list(send_gain))
            if use_pan:
                sc.set_pan(voices, pans)
        sc.mix(1024, post_process=hrtf)
        parts = [sc.dry().ravel()] + [sc.wet(s).ravel() for s in range(2)]
        outs.append(np.concatenate(parts).copy())
    sc.close()
    return outs


CASES = [("exact", 1, False, False), ("fast", 2, False, False), ("fast", 3, False, False), ("exact", 4, False, False),
         ("fast", 4, True, False), ("exact", 2, True, False), ("fast", 1, False, True), ("fast", 1, True, True)]


@pytest.mark.parametrize("mode,order,spread_on,hrtf", CASES)
def test_pan_on_gpu_equals_host_computed_gains(mode, order, spread_on, hrtf, synth_mhr):
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    host = run(mode, order, spread_on, hrtf, synth_mhr, use_pan=False)
    gpu = run(mode, order, spread_on, hrtf, synth_mhr, use_pan=True)
    for k, (a, b) in enumerate(zip(gpu, host)):
        assert np.abs(b).max() > 1e-3
        if spread_on:
            assert np.abs(a - b).max() <= 1e-6 * np.abs(b).max(), (k, float(np.abs(a - b).max()))
        else:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (k, float(np.abs(a - b).max()))


@pytest.mark.parametrize("mode,order,spread_on", [("fast", 2, False), ("fast", 3, True), ("exact", 2, True)])
def test_pan_on_gpu_against_an_oracle_mixed_scene(mode, order, spread_on, synth_mhr):
    """The direct comparison: the same voices mixed by the REFERENCE (Voice::mix of the compiled reference, or the pinned
    restatement) with line gains the reference's own CalcDirectionCoeffs / ComputePanGains arithmetic produced, against the
    GPU scene that was only told directions, spreads and gains (oalgpu_voice_set_pan).  Buses after every update, to the
    multi-voice figure of every other scene test."""
    import oalgpu
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    which = "ref" if ol.available("ref") else "port"
    L = ol.load(which)
    api = oalgpu.Api(oalgpu.MATH_EXACT if mode == "exact" else oalgpu.MATH_FAST)
    rng = np.random.default_rng(77 + order)
    ndry = min(ACN_COUNT[order], 24)
    nvoices, sends = 11, 2
    gsc = api.make_scene(num_dry=ndry, num_real=0, num_sends=sends, num_slots=2, wet_channels=4, hrtf=False, max_voices=nvoices)
    osc = ol.Scene(L, sample_rate=48000, num_dry=ndry, num_real=0, num_sends=sends, num_slots=2, wet_channels=4, hrtf=False)
    dry_idx = rng.permutation(ACN_COUNT[order])[:ndry].astype(np.uint8)
    dry_scale = rng.uniform(0.5, 1.5, ndry).astype(np.float32)
    wet_idx = [np.arange(4, dtype=np.uint8), np.array([0, 3, 1, 2], np.uint8)]
    wet_scale = [np.ones(4, np.float32), rng.uniform(0.5, 1.5, 4).astype(np.float32)]
    gsc.set_ambi_map(dry_idx, dry_scale)
    for s in range(2):
        gsc.set_slot_ambi_map(s, wet_idx[s], wet_scale[s])
    data = rng.uniform(-1, 1, 6000).astype(np.float32)
    gbuf = gsc.add_buffer(data, ol.FMT_FLOAT, loop_start=0, loop_end=6000)
    obuf = osc.add_buffer(data, ol.FMT_FLOAT, loop_start=0, loop_end=6000)
    for v in range(nvoices):
        gsc.add_voice(gbuf, looping=True, position=(v * 577) % 5000, frac=(v * 4001) % 65536)
        osc.add_voice(obuf, True, position=(v * 577) % 5000, frac=(v * 4001) % 65536)
    worst = 0.0
    for k in range(4):
        r = np.random.default_rng(500 * k + 3)
        voices, pans = [], []
        for v in range(nvoices):
            d = r.standard_normal(3)
            d = (d / np.linalg.norm(d)).astype(np.float32)
            spread = np.float32(r.uniform(0.1, 6.0)) if (spread_on and v % 2 == 0) else np.float32(0.0)
            dry_gain = np.float32(r.uniform(0.05, 0.6))
            send_gain = r.uniform(0.05, 0.5, 6).astype(np.float32)
            slots = [(v + i) % 3 - 1 for i in range(sends)]
            coeffs = L.direction_coeffs(d, float(spread))
            dry = (dry_scale * coeffs[dry_idx]) * dry_gain
            snd_o, snd_g = [], []
            for i in range(sends):
                g = np.zeros(4, np.float32)
                if slots[i] >= 0:
                    g = (wet_scale[slots[i]] * coeffs[wet_idx[slots[i]]]) * send_gain[i]
                filt = ol.default_filter(active=(v + i) % 2, gain_hf=0.6)
                snd_o.append((slots[i], g, filt))
                snd_g.append((slots[i], np.zeros(4, np.float32), filt))
            df = ol.default_filter(active=v % 2, gain_hf=0.5)
            osc.set_params(v, ol.make_voice_params(60211, ol.RS_BSINC24, dry_gains=dry, direct_filter=df, sends=snd_o))
            gsc.set_params(v, ol.make_voice_params(60211, ol.RS_BSINC24, dry_gains=np.zeros(ndry), direct_filter=df, sends=snd_g))
            voices.append(v)
            pans.append(list(d) + [spread, dry_gain] + list(send_gain))
        gsc.set_pan(voices, pans)
        gsc.mix(1024, post_process=False)
        osc.mix(1024, post_process=False)
        for name, got, want in [("dry", gsc.dry()[:ndry], osc.dry_view()[:ndry])] + [(f"wet{s}", gsc.wet(s), osc.wet(s)) for s in range(2)]:
            scale = float(np.abs(want).max())
            err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
            assert scale > 1e-3, (k, name)
            assert err <= 2e-5 * scale + 1e-7, (k, name, err, scale)
            worst = max(worst, err / scale)
    print(f"panning against the oracle-mixed scene ({mode}, order {order}, spread {spread_on}): worst {worst:.2e} of the bus maximum")
    gsc.close()
    osc.close()
