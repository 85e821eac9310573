"""Profiling aid: per-phase s_memtime deltas of the wavefront voice kernel's measurement variant
(contexts created with OALGPU_CTX_PROFILE | OALGPU_CTX_SERIAL; add 1 = OALGPU_CTX_FIR_VALU as argv[1])."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "measure"))
import numpy as np
import oalgpu
import oalmeasure
oalmeasure.use_measurement_build()      # liboalgpu_measure.so: the product's sources + the oalgpu_debug_* readers
from oalgpu import synth
import bench
V = 4096
api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=oalgpu.CTX_WAVE_PAIRS | oalgpu.CTX_PROFILE | oalgpu.CTX_SERIAL | (int(sys.argv[1]) if len(sys.argv) > 1 else 0))
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
names = ["src in LDS", "resample", "biquad", "hist+x' build", "request next", "FIR(+old pass)", "write-back"]
t2 = np.stack([t[:, 0], t[:, 7], t[:, 1], t[:, 2], t[:, 3], t[:, 4], t[:, 5], t[:, 6]], axis=1)
d = np.diff(t2, axis=1)
kinds = {"static unfiltered": [v for v in allv if v % 4 in (2, 3)], "filtered": [v for v in allv if v % 4 == 1], "moving": moving}
# (s_memtime counters of different XCDs do not share a time base: only differences within one
# wavefront are meaningful)
print("s_memtime ticks per phase, mean over voices:")
for kn, vs in kinds.items():
    print(kn, " ".join(f"{n}={d[vs, i].mean():.0f}" for i, n in enumerate(names)), "sum=%.0f" % d[vs].sum(axis=1).mean())

# per-wavefront view: wave w of a group of four voices mixes voices (w, w+2); span in ticks
tt = t.reshape(-1, 2, 2, 8)                    # [group][k][parity][stamp]; voice = 4*group + 2*k + parity
starts = np.minimum(tt[:, 0, :, 0], tt[:, 1, :, 0]); ends = np.maximum(tt[:, 0, :, 6], tt[:, 1, :, 6])     # [group][parity] (either order)
dur = ends - starts
print("wave busy ticks: mean=%.0f p50=%.0f p99=%.0f max=%.0f" % (dur.mean(), np.median(dur), np.percentile(dur, 99), dur.max()))
for par, nm in ((0, "moving+static"), (1, "filtered+static")):
    d_ = dur[:, par]
    print("  %s waves: mean=%.0f p50=%.0f p99=%.0f max=%.0f" % (nm, d_.mean(), np.median(d_), np.percentile(d_, 99), d_.max()))
order = [0, 7, 1, 2, 3, 4, 5, 6]
ph = np.diff(tt[..., order], axis=-1)          # [group][k][parity][7]
gap = np.abs(np.maximum(tt[:, 0, :, 0], tt[:, 1, :, 0]) - np.minimum(tt[:, 0, :, 6], tt[:, 1, :, 6]))   # between the two voices of a wave
slow = dur >= np.percentile(dur, 99)
print("phase means, all waves : first voice", " ".join("%.0f" % x for x in ph[:, 0].reshape(-1, 7).mean(axis=0)), "| gap %.0f |" % gap.mean(), "second", " ".join("%.0f" % x for x in ph[:, 1].reshape(-1, 7).mean(axis=0)))
print("phase means, slowest 1%: first voice", " ".join("%.0f" % x for x in ph[:, 0][slow].mean(axis=0)), "| gap %.0f |" % gap[slow].mean(), "second", " ".join("%.0f" % x for x in ph[:, 1][slow].mean(axis=0)))

# are slow wavefronts a workgroup / CU effect?  (workgroup = 8 voices = 4 waves)
dw = dur.reshape(-1, 4)                         # [workgroup][wave-ish]
wgmax = dw.max(axis=1); wgmin = dw.min(axis=1)
print("per-workgroup: mean of max=%.0f mean of min=%.0f; corr(max,min)=%.2f" % (wgmax.mean(), wgmin.mean(), np.corrcoef(wgmax, wgmin)[0, 1]))
idx = np.argsort(-wgmax)[:12]
print("slowest workgroups (index, index%8, wave durations):")
for i in idx: print("  ", i, i % 8, dw[i].tolist())
byx = [wgmax[i::8].mean() for i in range(8)]
print("mean wg max by index%8:", " ".join("%.0f" % x for x in byx))

half = dw.shape[0] // 2
print("workgroups of the first half: mean of max=%.0f max=%.0f; second half: mean of max=%.0f max=%.0f" % (wgmax[:half].mean(), wgmax[:half].max(), wgmax[half:].mean(), wgmax[half:].max()))

# per-wavefront stamps: entry, first voice requested+parked (pass 0), voices done, partial stored
wt = np.zeros((V, 8), np.uint64); nw = C.c_uint32(0)
rc = oalgpu.lib.oalgpu_debug_wave_times(sc.h, wt.ctypes.data_as(C.c_void_p), C.byref(nw)); assert rc == 0, rc
wt = wt[:nw.value].astype(np.int64)
d0 = wt[:, 1] - wt[:, 0]; d1 = wt[:, 2] - wt[:, 1]; d2 = wt[:, 3] - wt[:, 2]; tot = wt[:, 3] - wt[:, 0]
for nm, x in (("pass 0 (first head/buffer/window + table staging)", d0), ("voices", d1), ("dump + partial store", d2), ("wave lifetime", tot)):
    print("%-52s mean=%.0f p50=%.0f p99=%.0f max=%.0f" % (nm, x.mean(), np.median(x), np.percentile(x, 99), x.max()))
print("pass 0 in detail (cycles after entry): control line in registers %.0f, request issued %.0f, staging barrier passed %.0f, first voice parked %.0f"
      % ((wt[:, 4] - wt[:, 0]).mean(), (wt[:, 5] - wt[:, 0]).mean(), (wt[:, 6] - wt[:, 0]).mean(), d0.mean()))
print("  the first request begins at %.0f (accumulators cleared, the pass loop entered)" % (wt[:, 7] - wt[:, 0]).mean())
h2 = nw.value // 2
print("wave lifetime, first-half workgroups: mean=%.0f max=%.0f; second half: mean=%.0f max=%.0f" % (tot[:h2].mean(), tot[:h2].max(), tot[h2:].mean(), tot[h2:].max()))

# cold start: the same kind of voice (static, unfiltered) when it is the FIRST voice its wavefront mixes
# (second-half workgroups run their voices in reverse order) against when it is the second
ngroups = V // 8
grp = np.arange(V) // 8
is_static = np.isin(np.arange(V) % 4, (2, 3))
first_static = is_static & (grp >= (ngroups + 1) // 2)
second_static = is_static & (grp < (ngroups + 1) // 2)
print("static voice as the wave's FIRST voice :", " ".join(f"{n}={d[first_static, i].mean():.0f}" for i, n in enumerate(names)), "sum=%.0f" % d[first_static].sum(axis=1).mean())
print("static voice as the wave's SECOND voice:", " ".join(f"{n}={d[second_static, i].mean():.0f}" for i, n in enumerate(names)), "sum=%.0f" % d[second_static].sum(axis=1).mean())
