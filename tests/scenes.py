"""Small deterministic multi-voice scenes used by the oracle pin test, the golden-vector
generator and the GPU parity tests.  A scene is driven through any object exposing the
oracle_lib.Scene interface (add_buffer/add_voice/set_params/set_state/mix/dry/wet/...)."""
import numpy as np

import oracle_lib as ol


def run_scene(L, MHR, hrtf, fmt, resampler, steps, n_updates, nvoices, rng_seed, sends=0, todo=1024,
               nonloop=False, stop_at=None, move=True, kernel_names=None, distances=None):
    rng = np.random.default_rng(rng_seed)
    if hrtf:
        L.hrtf_load(MHR)
    sc = L.make_scene(num_dry=4 if hrtf else 5, num_real=2 if hrtf else 0, num_sends=sends,
                      num_slots=2 if sends else 0, wet_channels=4, hrtf=hrtf)
    if kernel_names is not None and hasattr(sc, "voice_kernel_name"):
        kernel_names.append(sc.voice_kernel_name())           # which product kernel mixes this scene
    if hrtf:
        cc = np.zeros((4, 128, 2), np.float32)
        cc[:, :64] = rng.uniform(-0.2, 0.2, (4, 64, 2))
        sc.set_direct_hrtf(cc, [1.0, 0.8, 0.8, 0.8], 400.0 / 48000.0, 64)
    bufs = []
    for b in range(4):
        if fmt == ol.FMT_FLOAT:
            d = rng.uniform(-1, 1, 6000).astype(np.float32)
        elif fmt == ol.FMT_SHORT:
            d = rng.integers(-32768, 32767, 6000).astype(np.int16)
        elif fmt == ol.FMT_DOUBLE:
            d = rng.uniform(-1, 1, 6000)
        elif fmt == ol.FMT_INT:
            d = rng.integers(-2 ** 31, 2 ** 31 - 1, 6000).astype(np.int32)
        else:
            d = rng.integers(0, 255, 6000).astype(np.uint8)
        bufs.append(sc.add_buffer(d, fmt, loop_start=100 * b, loop_end=6000 - 50 * b))

    def params(v, k):
        r = np.random.default_rng(rng_seed * 1000 + v * 17 + k)
        filt = ol.default_filter(active=1 if v % 4 == 1 else 0, gain_hf=0.5 if k < 2 else 0.2)
        snd = []
        for i in range(sends):
            sf = ol.default_filter(active=1 if (v + i) % 3 == 0 else 0, gain_hf=0.7, gain_lf=0.8)
            snd.append((i % 2 if (v + i) % 5 else -1, r.uniform(0, 0.3, 4), sf))
        if hrtf:
            return ol.make_voice_params(steps[v % len(steps)], resampler,
                                        hrtf=(np.arcsin(r.uniform(-1, 1)), r.uniform(-np.pi, np.pi),
                                              distances[(v + k) % len(distances)] if distances else 2.0, 0.0,
                                              10 ** (r.uniform(-60, -20) / 20)),
                                        direct_filter=filt, sends=snd)
        return ol.make_voice_params(steps[v % len(steps)], resampler, dry_gains=r.uniform(0, 0.1, 5),
                                    direct_filter=filt, sends=snd)

    for v in range(nvoices):
        sc.add_voice(bufs[v % 4], looping=not (nonloop and v % 2 == 0), position=(v * 7919) % 5000,
                     frac=(v * 977) % 65536)
        sc.set_params(v, params(v, 0))
    out = []
    for k in range(n_updates):
        if k > 0 and move:
            for v in range(0, nvoices, 4):
                sc.set_params(v, params(v, k))
        if stop_at is not None and k == stop_at:
            for v in range(1, nvoices, 3):
                sc.set_state(v, ol.VOICE_STOPPING)
        sc.mix(todo, post_process=hrtf)
        out.append(sc.dry().ravel())
        if hrtf:
            out.append(sc.hrtf_accum().ravel())
        for sl in range(2 if sends else 0):
            out.append(sc.wet(sl).ravel())
    ints = []
    for v in range(nvoices):
        st = sc.voice_state(v)
        ints.append((st.play_state, st.position, st.position_frac, st.has_buffer, st.fading,
                     st.hrtf_old_delay[0], st.hrtf_old_delay[1], st.direct_lp.counter))
        out.append(np.array(st.prev_samples, np.float32))
        out.append(np.array(st.dry_current, np.float32))
        out.append(np.array(st.hrtf_history, np.float32))
        out.append(np.array([st.hrtf_old_gain] + list(st.direct_lp.as_tuple()[:12])
                            + list(st.direct_hp.as_tuple()[:12]), np.float32))
        for i in range(sends):
            out.append(np.array(st.send_current[i], np.float32))
            out.append(np.array(st.send_lp[i].as_tuple()[:12], np.float32))
    sc.close()
    return np.concatenate(out), ints


SCENES = [
    dict(hrtf=False, fmt=ol.FMT_FLOAT, resampler=ol.RS_LINEAR, steps=[60211], n_updates=3, nvoices=8),
    dict(hrtf=False, fmt=ol.FMT_SHORT, resampler=ol.RS_BSINC24, steps=[60211, 65536, 40000],
         n_updates=4, nvoices=12, sends=2),
    dict(hrtf=False, fmt=ol.FMT_FLOAT, resampler=ol.RS_BSINC24, steps=[90000, 200000, 655360],
         n_updates=3, nvoices=9, sends=1),
    dict(hrtf=True, fmt=ol.FMT_FLOAT, resampler=ol.RS_BSINC24, steps=[60211], n_updates=4, nvoices=12),
    dict(hrtf=True, fmt=ol.FMT_UBYTE, resampler=ol.RS_SPLINE, steps=[60211, 131072], n_updates=3,
         nvoices=8, sends=2, todo=1000),
    dict(hrtf=False, fmt=ol.FMT_DOUBLE, resampler=ol.RS_GAUSSIAN, steps=[60211, 300000], n_updates=5,
         nvoices=8, nonloop=True),
    dict(hrtf=True, fmt=ol.FMT_INT, resampler=ol.RS_FAST_BSINC12, steps=[60211, 500000], n_updates=5,
         nvoices=8, nonloop=True, stop_at=2),
    dict(hrtf=False, fmt=ol.FMT_MULAW, resampler=ol.RS_BSINC48, steps=[70000], n_updates=2, nvoices=4),
    dict(hrtf=False, fmt=ol.FMT_ALAW, resampler=ol.RS_POINT, steps=[65536, 1], n_updates=2, nvoices=4,
         todo=37),
    # three sends, sends 0 and 2 into the SAME slot, filtered and unfiltered sends side by side,
    # voices stopping on the way: the stream-row path of the wavefront kernel
    dict(hrtf=False, fmt=ol.FMT_FLOAT, resampler=ol.RS_BSINC24, steps=[60211, 48000], n_updates=4, nvoices=40,
         sends=3, stop_at=1, todo=700),
    dict(hrtf=True, fmt=ol.FMT_SHORT, resampler=ol.RS_BSINC24, steps=[60211, 70000], n_updates=4, nvoices=36,
         sends=3, stop_at=2, nonloop=True),
]


