"""Where an update's period goes beyond the voice kernel: periods of back-to-back updates with and
without the parameter block, the post-process and the overlapped (two-stream) path."""
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
N = 400
def run(name, fn):
    for k in range(20): fn(k)
    sc.sync()
    t0 = time.perf_counter()
    for k in range(N): fn(k)
    t1 = time.perf_counter(); sc.sync(); t2 = time.perf_counter()
    print("%-46s host %.1f us  period %.1f us" % (name, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
run("apply + mix(post)", lambda k: (sc.apply_block(blocks[k % 40]), sc.mix(1024, post_process=True)))
run("mix(post)", lambda k: sc.mix(1024, post_process=True))
run("mix(no post: voices + reduce)", lambda k: sc.mix(1024, post_process=False))
run("serial entry points: mix_voices only", lambda k: sc.mix_voices(1024))
run("serial: mix_voices + post_process", lambda k: (sc.mix_voices(1024), sc.post_process(1024)))
run("apply only", lambda k: sc.apply_block(blocks[k % 40]))
run("mix(no post: voices + reduce) again", lambda k: sc.mix(1024, post_process=False))
run("mix(post) again", lambda k: sc.mix(1024, post_process=True))
run("apply + mix(post) again", lambda k: (sc.apply_block(blocks[k % 40]), sc.mix(1024, post_process=True)))
sc.set_timing(True)
run("serial mix_voices only, 3 event records per step", lambda k: sc.mix_voices(1024))
sc.set_timing(False)
run("serial mix_voices only", lambda k: sc.mix_voices(1024))
