"""Profiling aid: per-phase cycle-counter deltas of the reverb kernel's four role wavefronts
(oalgpu_reverb_debug_enable_phase_times): early = taps, biquad, all-pass, reflect+fence, delay/scatter;
late = mod+cubic taps, T60 biquad, wait for early, late-in add, vector all-pass, out+feedback."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openal-soft_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "measure"))
import oalgpu  # noqa: E402
import oalmeasure  # noqa: E402
oalmeasure.use_measurement_build()

rng = np.random.default_rng(0)
g = oalgpu.Reverb(4)
if len(sys.argv) > 1 and sys.argv[1] == "fast":
    g.set_math_mode(oalgpu.MATH_FAST)
assert oalgpu.lib.oalgpu_reverb_debug_enable_phase_times(g.h) == 0
g.update(oalgpu.ReverbProps.make(modulation_depth=0.5))
x = (rng.standard_normal((4, 1024)) * 0.1).astype(np.float32)
o = np.zeros((4, 1024), np.float32)
for k in range(20):
    g.process(x, o)
buf = (C.c_ulonglong * 256)()
assert oalgpu.lib.oalgpu_reverb_debug_phase_times(g.h, buf) == 0
t = np.array(buf, np.int64).reshape(4, 8, 8)
k0 = t[0, 7, 0]
print("kernel: phase0 %d, waves %d, mix %d ticks" % (t[0, 7, 1] - k0, t[0, 7, 2] - t[0, 7, 1], t[0, 7, 3] - t[0, 7, 2]))
names = {0: ["taps", "biquad", "allpass", "reflect", "delay+scatter"],
         1: ["cubic", "t60", "wait", "late-in", "vecap", "out+fb"]}
for role in (0, 1):
    for sub in range(4):
        row = t[role, sub]
        nm = names[role]
        d = [int(row[i + 1] - row[i]) for i in range(len(nm))]
        print("role %d sub %d start %7d: " % (role, sub, row[0] - k0) + "  ".join(f"{a}={b}" for a, b in zip(nm, d)))
g.close()
