"""Near-field control scenes (DoNfcMix, core/voice.cpp:904-932): mono voices panned
onto an ambisonic dry bus whose device has a control distance, so every order's lines are mixed
from that order's NFC-filtered copy of the voice (NfcFilter, core/filters/nfc.cpp).  Driven
through any object with the oracle_lib.Scene interface."""
import numpy as np

import oracle_lib as ol

SPEED_OF_SOUND = 343.3


def run(L, order=2, n_updates=4, todo=(1024, 1024, 500, 1024), seed=2, nvoices=7, sends=1, periphonic=True,
        resampler=ol.RS_BSINC24, nfc_every=1):
    rng = np.random.default_rng(seed)
    cpo = ([1, 3, 5, 7, 9] if periphonic else [1, 2, 2, 2, 2])[:order + 1]
    nlines = sum(cpo)
    sc = L.make_scene(num_dry=nlines, num_real=0, num_sends=sends, num_slots=1 if sends else 0, wet_channels=4,
                      hrtf=False)
    sc.set_nfc(SPEED_OF_SOUND / (1.5 * 48000.0), cpo)      # control distance 1.5 m
    buf = sc.add_buffer(rng.uniform(-1, 1, 7000).astype(np.float32), ol.FMT_FLOAT, loop_start=10, loop_end=6990)

    def params(v, k):
        r = np.random.default_rng(seed * 7919 + v * 31 + k)
        snd = [(0, r.uniform(0, 0.3, 4), ol.default_filter(active=v % 2, gain_hf=0.6)) for _ in range(sends)]
        return ol.make_voice_params([60211, 70000, 48000][v % 3], resampler, dry_gains=r.uniform(-0.2, 0.2, nlines),
                                    direct_filter=ol.default_filter(active=(v % 3 == 1), gain_hf=0.5), sends=snd)

    def w0(v, k):       # source distance 0.4 .. 4 m, moving for every other voice
        dist = 0.4 + 0.5 * v + (0.3 * k if v % 2 == 0 else 0.0)
        return SPEED_OF_SOUND / (max(dist, 1.5 / 4.0) * 48000.0)

    for v in range(nvoices):
        sc.add_voice(buf, looping=True, position=(v * 611) % 5000, frac=(v * 4001) % 65536)
        sc.set_params(v, params(v, 0))
        if v % nfc_every == 0:
            sc.set_voice_nfc(v, w0(v, 0))
    out = []
    for k in range(n_updates):
        if k > 0:
            for v in range(0, nvoices, 2):
                sc.set_params(v, params(v, k))
                if v % nfc_every == 0:
                    sc.set_voice_nfc(v, w0(v, k))
        if k == 2 and nvoices > 1:
            sc.set_state(1, ol.VOICE_STOPPING)
        n = todo[k % len(todo)]
        sc.mix(n, post_process=False)
        out.append(sc.dry()[:, :n].ravel())
        if sends:
            out.append(sc.wet(0)[:, :n].ravel())
    sc.close()
    return np.concatenate(out)
