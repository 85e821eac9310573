import os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, ROOT)
import torch, oalgpu
from oalgpu import synth
import bench
api = oalgpu.Api(oalgpu.MATH_FAST, device=0)
mhr = synth.synth_mhr_bytes(); api._mhr = mhr
V = 4096
sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
allv = list(range(V)); moving = [v for v in allv if script.is_moving(v)]
sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
blocks = [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(40)]
for k in range(10):
    sc.apply_block(blocks[k]); sc.mix(1024, post_process=True)
sc.sync()
N = 400
t0 = time.perf_counter()
for k in range(N):
    sc.apply_block(blocks[k % 40]); sc.mix(1024, post_process=True)
t1 = time.perf_counter()
sc.sync()
t2 = time.perf_counter()
print("host submit per step: %.1f us; total per step incl. drain: %.1f us" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
# only mix
t0 = time.perf_counter()
for k in range(N):
    sc.mix(1024, post_process=True)
t1 = time.perf_counter(); sc.sync(); t2 = time.perf_counter()
print("mix only: host %.1f us, total %.1f us" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
def timed(label, fn, n=400):
    sc.sync(); t0 = time.perf_counter()
    for k in range(n): fn(k)
    t1 = time.perf_counter(); sc.sync(); t2 = time.perf_counter()
    print("%-44s host %.1f us, total %.1f us per call" % (label, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
timed("apply_block only", lambda k: sc.apply_block(blocks[k % 40]))
timed("mix(post_process=False) only", lambda k: sc.mix(1024, post_process=False))
timed("mix(post_process=True) only", lambda k: sc.mix(1024, post_process=True))
timed("apply + mix(post_process=True)", lambda k: (sc.apply_block(blocks[k % 40]), sc.mix(1024, post_process=True)))
timed("apply + mix(post_process=False)", lambda k: (sc.apply_block(blocks[k % 40]), sc.mix(1024, post_process=False)))
