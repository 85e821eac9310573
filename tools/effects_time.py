#!/usr/bin/env python3
"""Kernel time of every oalgpu_effect kind for one 1024-sample block (run under rocprofv3 --kernel-trace --stats;
tools/r2_effects.sh).  Each effect is updated with typical EFX properties and processes 40 blocks of noise."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "openal-soft_amd"))
import oalgpu  # noqa: E402

PROPS = {
    oalgpu.EFFECT_EQUALIZER: [200.0, 2.0, 500.0, 0.5, 1.0, 3000.0, 3.0, 0.7, 6000.0, 0.3],
    oalgpu.EFFECT_MODULATOR: [440.0, 800.0, 0],
    oalgpu.EFFECT_ECHO: [0.1, 0.1, 0.5, 0.5, -1.0],
    oalgpu.EFFECT_COMPRESSOR: [1],
    oalgpu.EFFECT_CHORUS: [1, 90, 1.1, 0.1, 0.25, 0.016],
    oalgpu.EFFECT_DISTORTION: [0.2, 0.05, 8000.0, 3600.0, 3600.0],
    oalgpu.EFFECT_AUTOWAH: [0.06, 0.06, 1000.0, 11.22],
    oalgpu.EFFECT_VMORPHER: [1.41, 0, 10, 0, 0, 0],
    oalgpu.EFFECT_FSHIFTER: [100.0, 0, 1],
    oalgpu.EFFECT_PSHIFTER: [12, 0],
}

rng = np.random.default_rng(1)
x = (rng.standard_normal((4, 1024)) * 0.25).astype(np.float32)
for kind, props in PROPS.items():
    fx = oalgpu.Effect(kind, 4, 4, 48000, oalgpu.MATH_FAST)
    if kind == oalgpu.EFFECT_ECHO:
        fx.update(props, None, np.full((2, 4), 0.5, np.float32))
    else:
        fx.update(props, np.arange(4, dtype=np.uint32), np.ones(4, np.float32))
    out = np.zeros((4, 1024), np.float32)
    for _ in range(40):
        fx.process(x, out, 1024)
    fx.close()
print("done")
