"""B-Format (ambisonic) source scenes: a few 4-channel first-order voices (VoiceFlag::IsAmbisonic,
per-channel BandSplitter::processScale ahead of DoFilters, core/voice.cpp:1082-1091) next to
mono voices, mixed onto a 9-line (second-order) dry bus with one send.  Driven through any
object with the oracle_lib.Scene interface; the product's Scene implements the same calls."""
import numpy as np

import oracle_lib as ol

NLINES = 9
XOVER = 400.0 / 48000.0
# HF scales of a first-order source on a second-order device, per channel order (W, then XYZ):
# the numbers only need to be the same on both sides
HF_SCALES = (1.26, 0.91, 0.91, 0.91)


def run(L, n_updates=4, todo=(1024, 1024, 600, 1024), seed=3, sends=1, nambi=3, nmono=5, stop_at=2,
        resampler=ol.RS_BSINC24, only_channel=None):
    rng = np.random.default_rng(seed)
    sc = L.make_scene(num_dry=NLINES, num_real=0, num_sends=sends, num_slots=2 if sends else 0, wet_channels=4,
                      hrtf=False)
    bfmt = sc.add_buffer(rng.uniform(-1, 1, 4 * 7000).astype(np.float32), ol.FMT_FLOAT, frame_step=4,
                         loop_start=50, loop_end=6900)
    bshort = sc.add_buffer(rng.integers(-30000, 30000, 4 * 5000).astype(np.int16), ol.FMT_SHORT, frame_step=4)
    mono = sc.add_buffer(rng.uniform(-1, 1, 6000).astype(np.float32), ol.FMT_FLOAT)

    def params(key, k, filt_active):
        r = np.random.default_rng(seed * 7919 + key * 31 + k)
        snd = [(i % 2, r.uniform(0, 0.3, 4), ol.default_filter(active=(key + i) % 2, gain_hf=0.6)) for i in range(sends)]
        return ol.make_voice_params([60211, 70000, 48000][key % 3] if key < 100 else 60211, resampler,
                                    dry_gains=r.uniform(-0.2, 0.2, NLINES),
                                    direct_filter=ol.default_filter(active=filt_active, gain_hf=0.5, gain_lf=0.8),
                                    sends=snd)

    ambi = []
    for a in range(nambi):
        v = sc.add_ambi_voice(bfmt if a % 2 == 0 else bshort, 4, looping=(a % 2 == 0), position=(a * 977) % 3000,
                              frac=(a * 12345) % 65536)
        ambi.append(v)
        for c in range(4):
            # voice-wide fields (step, resampler, send slots, filter-active flags) identical per channel
            sc.set_channel_params(v, c, _chan_params(params, a, c, 0, only_channel))
            sc.set_channel_ambi_scale(v, c, XOVER, HF_SCALES[c], 1.0 if a != 1 else 0.7)
    monos = []
    for m in range(nmono):
        v = sc.add_voice(mono, looping=True, position=(m * 611) % 5000, frac=0)
        monos.append(v)
        sc.set_params(v, params(100 + m, 0, m % 2))
    out = []
    for k in range(n_updates):
        if k > 0:
            for a in range(0, nambi, 2):
                for c in range(4):
                    sc.set_channel_params(ambi[a], c, _chan_params(params, a, c, k, only_channel))
            if monos:
                sc.set_params(monos[0], params(100, k, 0))
        if stop_at is not None and k == stop_at and len(monos) > 1:
            sc.set_state(monos[1], ol.VOICE_STOPPING)
        n = todo[k % len(todo)]
        sc.mix(n, post_process=False)
        out.append(sc.dry()[:, :n].ravel())
        for sl in range(2 if sends else 0):
            out.append(sc.wet(sl)[:, :n].ravel())
    sc.close()
    return np.concatenate(out)


def _chan_params(params, a, c, k, only_channel=None):
    """Per-channel gains differ; everything voice-wide is taken from the voice's key.
    only_channel: every other channel gets zero gains (its bus contribution is an exact 0)."""
    p = params(a, k, a % 2)
    r = np.random.default_rng(1000 * a + 10 * c + k)
    g = r.uniform(-0.3, 0.3, NLINES).astype(np.float32)
    if only_channel is not None and c != only_channel:
        g[:] = 0.0
    for i in range(NLINES):
        p.dry_gains[i] = float(g[i])
    for s in range(6):
        for i in range(4):
            p.send_gains[s][i] = float(p.send_gains[s][i]) * (0.5 + 0.1 * c)
    return p
