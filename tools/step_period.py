"""Measurement aid: period of the config-3 step loop (parameter block + oalgpu_mix_update) with and without the HRTF post-process
on the post stream -- whether the post chain (reduction + post-process, beside the next update's voice kernel) or the main
chain bounds the step.  python tools/step_period.py [xflags [config [voices_per_workgroup [quick]]]]"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oalgpu
from oalgpu import synth
import bench
xf = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
V = 8192 if cfg == 4 else 4096
vpg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
quick = len(sys.argv) > 4
api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=xf)
mhr = open(os.path.join(ROOT, "tests", "golden", "default_hrtf.mhr"), "rb").read(); api._mhr = mhr
sc, script = bench.build_scene(oalgpu, synth, api, cfg, V, 0, mhr, vpg)
allv = list(range(V)); moving = [v for v in allv if script.is_moving(v)]
sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
blocks = [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(48)]
gc.collect(); gc.disable()
def run(n, post, params=True):
    for k in range(n):
        if params: sc.apply_block(blocks[k % len(blocks)])
        sc.mix(1024, post_process=post)
        if k % 25 == 24 and n <= 2000 and False: sc.sync()
def period(post, params=True, n=1000):
    run(1500, post, params); sc.sync()
    out = []
    for _ in range(3):
        t0 = time.perf_counter(); run(n, post, params); sc.sync()
        out.append((time.perf_counter() - t0) / n * 1e6)
    return " ".join("%.2f" % x for x in out)
print("xflags", xf, "config", cfg, "voices per workgroup", vpg or "auto")
print("params + mix, post-process on :", period(True))
print("params + mix, post-process off:", period(False))
if not quick:
    print("mix only,     post-process on :", period(True, False))
    print("mix only,     post-process off:", period(False, False))
