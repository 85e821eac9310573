"""Profiling aid: per-phase s_memtime deltas of the wavefront voice kernel's measurement variant on the dry-line and
send contexts (BASELINE configs 2, 4, 5): python tools/phase_times_lines.py CONFIG [VOICES]."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "measure"))
import numpy as np
import oalgpu
import oalmeasure
oalmeasure.use_measurement_build()      # liboalgpu_measure.so: the product's sources + the oalgpu_debug_* readers
from oalgpu import synth
import bench
config = int(sys.argv[1]) if len(sys.argv) > 1 else 2
V = int(sys.argv[2]) if len(sys.argv) > 2 else (8192 if config == 4 else 4096)
# (config 4's default form has its own reader, tools/phase_times_rows.py: this one reads the stream-row kernel's stamps)
api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=oalgpu.CTX_PROFILE | oalgpu.CTX_SERIAL | (oalgpu.CTX_STREAM_ROWS if config == 4 else 0))
mhr = synth.synth_mhr_bytes(); api._mhr = mhr
sc, script = bench.build_scene(oalgpu, synth, api, config, V, 0, mhr, 0)
allv = list(range(V)); moving = [v for v in allv if script.is_moving(v)]
sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
for k in range(6):
    sc.set_params_batch(moving, bench.param_array(oalgpu, script, moving, k + 1))
    sc.mix(1024, post_process=True)
sc.sync()
print("kernel:", sc.voice_kernel_name())
out = np.zeros((V, 8), np.uint64)
rc = oalgpu.lib.oalgpu_debug_phase_times(sc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0, rc
t = out.astype(np.int64)
names = ["src in LDS", "resample", "filters+rows", "hist+x' build", "request next", "FIR/park", "write-back"]
t2 = np.stack([t[:, 0], t[:, 7], t[:, 1], t[:, 2], t[:, 3], t[:, 4], t[:, 5], t[:, 6]], axis=1)
d = np.diff(t2, axis=1)
kinds = {"static unfiltered": [v for v in allv if v % 4 in (2, 3)], "filtered": [v for v in allv if v % 4 == 1], "moving": moving}
print("s_memtime ticks per phase, mean over voices:")
for kn, vs in kinds.items():
    print(" ", kn, " ".join(f"{n}={d[vs, i].mean():.0f}" for i, n in enumerate(names)), "sum=%.0f" % d[vs].sum(axis=1).mean())
if config == 4:
    for ns in range(5):
        vs = [v for v in allv if v % 5 == ns]
        print(f"  {ns} sends:", " ".join(f"{n}={d[vs, i].mean():.0f}" for i, n in enumerate(names)), "sum=%.0f" % d[vs].sum(axis=1).mean())
wt = np.zeros((V, 8), np.uint64); nw = C.c_uint32(0)
rc = oalgpu.lib.oalgpu_debug_wave_times(sc.h, wt.ctypes.data_as(C.c_void_p), C.byref(nw)); assert rc == 0, rc
wt = wt[:nw.value].astype(np.int64)
d0 = wt[:, 1] - wt[:, 0]; d1 = wt[:, 2] - wt[:, 1]; d2 = wt[:, 3] - wt[:, 2]; tot = wt[:, 3] - wt[:, 0]
for nm, x in (("pass 0 (first head/buffer/window + table staging)", d0), ("voices", d1), ("tail: dump / row mix + partial store", d2), ("wave lifetime", tot)):
    print("%-52s mean=%.0f p50=%.0f p99=%.0f max=%.0f" % (nm, x.mean(), np.median(x), np.percentile(x, 99), x.max()))
