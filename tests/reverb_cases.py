"""Shared EAX-reverb scenarios: a list of (name, schedule) where a schedule is a list of
per-update steps {props: kwargs for ReverbProps.make, slot_gain, n} -- None props = no update()
before that process() call.  Input is seeded noise bursts on the 4-line wet bus."""
import zlib

import numpy as np

BUFFER_LINE = 1024


def out_init(nlines):
    """process() ADDS into the target lines: start them from a recognisable non-zero pattern."""
    o = np.zeros((nlines, BUFFER_LINE), np.float32)
    o[:, :7] = 0.125
    return o


def wet_input(seed, updates, burst_every=3):
    rng = np.random.default_rng(seed)
    x = np.zeros((updates, 4, BUFFER_LINE), np.float32)
    for u in range(updates):
        if u % burst_every == 0:
            x[u] = (rng.standard_normal((4, BUFFER_LINE)) * 0.25).astype(np.float32)
            x[u, 1:] *= 0.5
    return x


def _steps(first, count, later=None):
    s = [dict(props=first, slot_gain=1.0, n=BUFFER_LINE)]
    s += [dict(props=None, slot_gain=1.0, n=BUFFER_LINE) for _ in range(count - 1)]
    for k, v in (later or {}).items():
        s[k] = dict(props=v, slot_gain=s[k].get("slot_gain", 1.0), n=BUFFER_LINE)
    return s


CASES = [
    # AL_EAXREVERB_DEFAULT_*: first update from DeviceClear, then steady state
    ("default", _steps({}, 8)),
    # modulation active (Mod.Depth > 0 -> fractional cubic taps on the feedback lines)
    ("modulated", _steps(dict(modulation_depth=0.8, modulation_time=0.7, diffusion=0.6), 8)),
    # panned early/late reflections and non-unit shelf gains
    ("panned", _steps(dict(reflections_pan=(0.3, 0.1, -0.4), late_reverb_pan=(-0.2, 0.0, 0.9), gain_hf=0.4,
                          gain_lf=0.7, reflections_gain=0.6, late_reverb_gain=2.0), 6)),
    # small density -> short lines (sub-blocks bounded by mLate.Offset[0] / VecAp.Offset[0])
    ("dense_small_room", _steps(dict(density=0.0, diffusion=0.3, decay_time=0.4, decay_hf_ratio=0.3,
                                     decay_lf_ratio=1.5, decay_hf_limit=0), 6)),
    # partial update (gain/pan/delays only: taps cross-fade, no pipeline swap)
    ("partial_update", _steps({}, 8, {3: dict(gain=0.6, reflections_delay=0.02, late_reverb_delay=0.03,
                                             reflections_pan=(0.5, 0.0, 0.0))})),
    # full update -> StartFade / Fading / Cleanup / Normal with both pipelines running
    ("pipeline_fade", _steps({}, 14, {2: dict(density=0.5, decay_time=2.5, diffusion=0.8)})),
    # two full updates in a row while the first fade is still running
    ("double_fade", _steps(dict(decay_time=0.3), 12, {2: dict(decay_time=3.0, modulation_depth=0.5),
                                                      4: dict(decay_time=1.0, density=0.2)})),
    # short / ragged process() sizes
    ("ragged", [dict(props={}, slot_gain=0.8, n=1024), dict(props=None, slot_gain=0.8, n=100),
                dict(props=None, slot_gain=0.8, n=1), dict(props=dict(decay_time=2.0), slot_gain=0.8, n=333),
                dict(props=None, slot_gain=0.8, n=1024), dict(props=None, slot_gain=0.8, n=257),
                dict(props=None, slot_gain=0.8, n=1024), dict(props=None, slot_gain=0.8, n=1024)]),
]

SEED = {name: zlib.crc32(name.encode()) % 1000 for name, _ in CASES}
FULL_CASES = ("default", "ragged")      # cases whose golden fixture holds the lines, not only CRCs
