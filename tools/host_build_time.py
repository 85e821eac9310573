import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, ROOT)
import oalgpu
from oalgpu import synth
import bench
api = oalgpu.Api(oalgpu.MATH_FAST, device=0)
mhr = synth.synth_mhr_bytes(); api._mhr = mhr
V = 4096
sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
allv = list(range(V)); moving = [v for v in allv if script.is_moving(v)]
sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
recs = [bench.param_array(oalgpu, script, moving, k + 1) for k in range(8)]
one = bench.param_array(oalgpu, script, moving[:1], 1)
for n, (vs, arr) in (("1024", (moving, recs)), ("1", (moving[:1], [one]))):
    for k in range(5): sc.set_params_batch(vs, arr[k % len(arr)])
    t0 = time.perf_counter()
    for k in range(200): sc.set_params_batch(vs, arr[k % len(arr)])
    print("set_params_batch of", n, "records: %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
t0 = time.perf_counter()
for k in range(200): sc.mix(1024, post_process=True); sc.dry()
print("mix + read_dry: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
