"""Profiling aid: s_memtime stamps of VoiceRowsKernel's measurement variant (OALGPU_CTX_ROW_SLICES | OALGPU_CTX_PROFILE | OALGPU_CTX_SERIAL),
per workgroup, round and wavefront: 0 round start, 1 window parked and resampled, 2 next request issued + the voice's unfiltered signal
resolved and published (in front of the round's first barrier), 3 behind that barrier, 4 the eight unfiltered rows consumed, 5 the round's
filter jobs posted, executed and consumed, 6 state written back, 7 behind the round's last barrier."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "measure"))
import numpy as np
import oalgpu
import oalmeasure
oalmeasure.use_measurement_build()
from oalgpu import synth
import bench
V = 8192
api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=oalgpu.CTX_ROW_SLICES | oalgpu.CTX_PROFILE | oalgpu.CTX_SERIAL)
mhr = synth.synth_mhr_bytes(); api._mhr = mhr
sc, script = bench.build_scene(oalgpu, synth, api, 4, V, 0, mhr, 0)
print("kernel:", sc.voice_kernel_name())
allv = list(range(V)); moving = [v for v in allv if script.is_moving(v)]
sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
for k in range(6):
    sc.set_params_batch(moving, bench.param_array(oalgpu, script, moving, k + 1))
    sc.mix(1024, post_process=True)
sc.sync()
a = np.zeros((V, 8), np.uint64); b = np.zeros((V, 8), np.uint64); nw = C.c_uint32(0)
assert oalgpu.lib.oalgpu_debug_phase_times(sc.h, a.ctypes.data_as(C.c_void_p)) == 0
assert oalgpu.lib.oalgpu_debug_wave_times(sc.h, b.ctypes.data_as(C.c_void_p), C.byref(nw)) == 0
t = np.concatenate([a.reshape(-1), b.reshape(-1)]).astype(np.int64).reshape(-1, 8, 8, 8)     # [group][round][wave][stamp]
G = int((t[:, 0, 0, 0] != 0).sum())
t = t[:G]
names = ["park+resample", "request+resolve", "barrier 1 wait", "consume A", "filter jobs", "write-back", "barrier 2 wait"]
for r in range(5):
    x = t[:, r]
    ok = x[:, :, 0] != 0
    if not ok.any(): continue
    d = np.diff(x, axis=2)
    print("round", r, " ".join("%s=%.0f" % (n, d[:, :, i][ok].mean()) for i, n in enumerate(names)), "| round (wave 0, start to end) %.0f" % (x[:, 0, 7] - x[:, 0, 0])[ok[:, 0]].mean())
life = t[:, :, 0, 7].max(axis=1) - t[:, 0, 0, 0]
print("workgroup, first round start to last round end: mean %.0f max %.0f ticks" % (life.mean(), life.max()))
