"""Profiling aid: per-phase s_memtime deltas of the workgroup-per-voice kernel (OALGPU_PHASE_TIMES=1)."""
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
vpg = int(os.environ.get("VPG", "0"))
api = oalgpu.Api(oalgpu.MATH_FAST)
mhr = synth.synth_mhr_bytes(); api._mhr = mhr
sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, vpg)
print(sc.voice_kernel_name())
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
names = ["park+barrier", "resample", "filters", "x' build", "request+FIR+fold", "old pass+write-back", "end barrier"]
d = np.diff(t, axis=1)
kinds = {"static unfiltered": [v for v in allv if v % 4 in (2, 3)], "filtered": [v for v in allv if v % 4 == 1], "moving": moving}
print("s_memtime ticks per phase, mean over voices:")
for kn, vs in kinds.items():
    print(kn, " ".join(f"{n}={d[vs, i].mean():.0f}" for i, n in enumerate(names)), "sum=%.0f" % d[vs].sum(axis=1).mean())
per = max(1, vpg) if vpg else max(1, (V + 1023) // 1024)
g = t.reshape(-1, per, 8)
life = g[:, -1, 7] - g[:, 0, 0]
print("workgroup voices span: mean=%.0f p50=%.0f p99=%.0f max=%.0f" % (life.mean(), np.median(life), np.percentile(life, 99), life.max()))
if per > 1:
    gaps = g[:, 1:, 0] - g[:, :-1, 7]
    print("gap between voices: mean=%.0f" % gaps.mean())
    firstv = d[::per].sum(axis=1).mean(); rest = np.concatenate([d[i::per] for i in range(1, per)]).sum(axis=1).mean()
    print("first voice of a workgroup: %.0f ticks, later voices: %.0f" % (firstv, rest))
    print("park+barrier: first %.0f, later %.0f" % (d[::per, 0].mean(), np.concatenate([d[i::per, 0] for i in range(1, per)]).mean()))
