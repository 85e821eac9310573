"""Measurement aid: what the HRTF post-process chain of a config-3 context costs ALONE on the GPU (nothing beside it) --
periods of back-to-back launches of (a) the fused post-process, (b) reduction + fused post-process (oalgpu_mix_voices without
voices is not available, so (b) is measured as mix_voices + post_process minus mix_voices alone)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oalgpu
from oalgpu import synth
import bench
V = 4096
api = oalgpu.Api(oalgpu.MATH_FAST)
mhr = synth.synth_mhr_bytes(); api._mhr = mhr
sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
allv = list(range(V))
sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
sc.mix_voices(1024); sc.post_process(1024); sc.sync()
def period(fn, n=400):
    for _ in range(20): fn()
    sc.sync()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    sc.sync()
    return (time.perf_counter() - t0) / n * 1e6
print("fused post-process alone, back to back: %.2f us per launch" % period(lambda: sc.post_process(1024)))
a = period(lambda: sc.mix_voices(1024), 200)
b = period(lambda: (sc.mix_voices(1024), sc.post_process(1024)), 200)
print("voice kernel + reduction: %.2f us; + fused post-process: %.2f us (difference %.2f)" % (a, b, b - a))
