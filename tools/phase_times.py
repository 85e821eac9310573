"""Profiling aid: per-phase s_memtime deltas of the wavefront voice kernel (OALGPU_PHASE_TIMES=1)."""
import os, sys, ctypes as C
os.environ["OALGPU_PHASE_TIMES"] = "1"
os.environ.setdefault("OALGPU_SERIAL", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import oalgpu
from oalgpu import synth
import bench
V = 4096
api = oalgpu.Api(oalgpu.MATH_FAST)
mhr = synth.synth_mhr_bytes(); api._mhr = mhr
sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
allv = list(range(V)); moving = [v for v in allv if script.is_moving(v)]
sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
for k in range(6):
    sc.set_params_batch(moving, bench.param_array(oalgpu, script, moving, k + 1))
    sc.mix(1024, post_process=True)
sc.sync()
out = np.zeros((V, 8), np.uint64)
oalgpu.lib.oalgpu_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
rc = oalgpu.lib.oalgpu_debug_phase_times(sc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0, rc
t = out.astype(np.int64)
names = ["src in LDS", "resample", "biquad", "hist+x' build", "request next", "FIR(+old pass)", "write-back"]
t2 = np.stack([t[:, 0], t[:, 7], t[:, 1], t[:, 2], t[:, 3], t[:, 4], t[:, 5], t[:, 6]], axis=1)
d = np.diff(t2, axis=1)
kinds = {"static unfiltered": [v for v in allv if v % 4 in (2, 3)], "filtered": [v for v in allv if v % 4 == 1], "moving": moving}
print("s_memtime ticks per phase (mean over voices), total kernel span %d ticks" % (t[:, 6].max() - t[:, 0].min()))
for kn, vs in kinds.items():
    print(kn, " ".join(f"{n}={d[vs, i].mean():.0f}" for i, n in enumerate(names)), "sum=%.0f" % d[vs].sum(axis=1).mean())
# first vs second voice of a wave
first = [v for v in allv if v % 2 == 0]; second = [v for v in allv if v % 2 == 1]
for kn, vs in (("first voice of wave", first), ("second voice of wave", second)):
    print(kn, " ".join(f"{n}={d[vs, i].mean():.0f}" for i, n in enumerate(names)), "start-offset=%.0f" % (t[vs, 0] - t[:, 0].min()).mean())
