"""Profiling aid: runs the EAX reverb kernel in its two regimes -- steady state (one pipeline)
and cross-fading after a full update (two pipelines) -- for rocprofv3 --kernel-trace --stats."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openal-soft_amd"))
import oalgpu  # noqa: E402

rng = np.random.default_rng(0)
g = oalgpu.Reverb(4)
g.update(oalgpu.ReverbProps.make(modulation_depth=0.5))
x = (rng.standard_normal((4, 1024)) * 0.1).astype(np.float32)
o = np.zeros((4, 1024), np.float32)
reps = int(os.environ.get("REPS", "200"))
mode = os.environ.get("MODE", "steady")
for k in range(reps):
    if mode == "fade" and k % 20 == 0:
        g.update(oalgpu.ReverbProps.make(modulation_depth=0.5, decay_time=1.0 + 0.01 * (k % 7), late_reverb_gain=10.0,
                                         gain=1.0))
    g.process(x, o)
g.close()
print("done", mode, reps)
