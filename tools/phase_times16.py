"""Profiling aid: per-phase s_memtime deltas of VoiceWave16Kernel's measurement variant (contexts created with
OALGPU_CTX_WAVE16 | OALGPU_CTX_PROFILE | OALGPU_CTX_SERIAL).  Stamps per voice: 0 kernel entry, 7 rows staged + window parked
(the workgroup's barrier passed), 1 resampled, 2 filtered, 3 ear 0 (x' build + FIR), 4 ear 1, 5 state written back, 6 dump +
partial stored."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "measure"))
import numpy as np
import oalgpu
import oalmeasure
oalmeasure.use_measurement_build()      # liboalgpu_measure.so: the product's sources + the oalgpu_debug_* readers
from oalgpu import synth
import bench
V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=oalgpu.CTX_PROFILE | oalgpu.CTX_SERIAL)
mhr = synth.synth_mhr_bytes(); api._mhr = mhr
sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
allv = list(range(V)); moving = [v for v in allv if script.is_moving(v)]
sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
for k in range(6):
    sc.set_params_batch(moving, bench.param_array(oalgpu, script, moving, k + 1))
    sc.mix(1024, post_process=True)
sc.sync()
out = np.zeros((V, 8), np.uint64)
rc = oalgpu.lib.oalgpu_debug_phase_times(sc.h, out.ctypes.data_as(C.c_void_p)); assert rc == 0, rc
t = out.astype(np.int64)
out2 = np.zeros((V, 8), np.uint64); nw = C.c_uint32(0)
rc = oalgpu.lib.oalgpu_debug_wave_times(sc.h, out2.ctypes.data_as(C.c_void_p), C.byref(nw)); assert rc == 0, rc
t_in = out2.astype(np.int64)
names = ["entry->barrier", "resample", "filters", "ear0 x'+FIR", "ear1 x'+FIR", "write-back", "dump+partial"]
t2 = np.stack([t[:, 0], t[:, 7], t[:, 1], t[:, 2], t[:, 3], t[:, 4], t[:, 5], t[:, 6]], axis=1)
d = np.diff(t2, axis=1)
kinds = {"static unfiltered": [v for v in allv if v % 4 in (2, 3)], "filtered": [v for v in allv if v % 4 == 1], "moving": moving}
print("s_memtime ticks per phase, mean over voices (one voice per wavefront):")
for kn, vs in kinds.items():
    print(" ", kn, " ".join(f"{n}={d[vs, i].mean():.0f}" for i, n in enumerate(names)), "lifetime=%.0f" % (t[vs, 6] - t[vs, 0]).mean())
life = t[:, 6] - t[:, 0]
print("wave lifetime: mean=%.0f p50=%.0f p99=%.0f max=%.0f" % (life.mean(), np.median(life), np.percentile(life, 99), life.max()))
wg = life.reshape(-1, 16)
print("per workgroup: mean of max=%.0f, mean of min=%.0f" % (wg.max(axis=1).mean(), wg.min(axis=1).mean()))
# what the dump's barrier waits for: per workgroup, the last wavefront to reach it, and what follows
vv = np.arange(V).reshape(-1, 16)
rel = t - t[:, 0:1]                         # stamps relative to the wavefront's own entry
arrive = rel[:, 5].reshape(-1, 16)
print("arrival at the dump (ticks after entry): mean=%.0f; per workgroup last=%.0f first=%.0f" % (arrive.mean(), arrive.max(axis=1).mean(), arrive.min(axis=1).mean()))
print("from the workgroup's last arrival to the partial stored: %.0f" % (rel[:, 6].reshape(-1, 16).max(axis=1) - arrive.max(axis=1)).mean())
for i, n in enumerate(names):
    x = d[:, i].reshape(-1, 16)
    print("  %-14s per workgroup: mean=%.0f max=%.0f min=%.0f" % (n, x.mean(), x.max(axis=1).mean(), x.min(axis=1).mean()))
late = np.argmax(arrive, axis=1)
print("which slot arrives last (wave index within the workgroup, histogram):", np.bincount(late, minlength=16).tolist())
dl = d.reshape(-1, 16, 7)[np.arange(late.size), late]
print("the workgroup's LAST wavefront, phase by phase:", " ".join(f"{n}={dl[:, i].mean():.0f}" for i, n in enumerate(names)))
first = np.argmin(arrive, axis=1)
df = d.reshape(-1, 16, 7)[np.arange(first.size), first]
print("the workgroup's FIRST wavefront, phase by phase:", " ".join(f"{n}={df[:, i].mean():.0f}" for i, n in enumerate(names)))
g = 37
base = t[vv[g], 0].min()
print("workgroup", g, ": wave, voice, stamps relative to the first entry [entry barrier resampled filtered ear0 ear1 written dumped]")
order = [0, 7, 1, 2, 3, 4, 5, 6]
inv = {}
for w in range(16):
    a, b = w >> 2, w & 3
    inv[w] = g * 16 + 4 * a + ((a + b) & 3)
for w in range(16):
    v = inv[w]
    print("  w%2d v%%4=%d " % (w, v % 4), " ".join("%6d" % (t[v, k] - base) for k in order))

# the inside of the first ear's pass (stamps 8..11 of the voice): setup (history, gains) | x' build | response staged | FIR
sub = np.stack([t[:, 2], t_in[:, 0], t_in[:, 1], t_in[:, 2], t_in[:, 3], t[:, 3]], axis=1)
ds = np.diff(sub, axis=1)
for kn, vs in kinds.items():
    print("  ear 0 of", kn, ": setup=%.0f x'build=%.0f response=%.0f FIR=%.0f old pass/rest=%.0f" % tuple(ds[vs].mean(axis=0)))
