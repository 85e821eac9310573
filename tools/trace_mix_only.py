"""300 updates without parameter blocks (oalgpu_mix_update only), for a kernel trace of the two streams."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, ROOT)
import oalgpu
from oalgpu import synth
import bench
api = oalgpu.Api(oalgpu.MATH_FAST, device=0)
mhr = synth.synth_mhr_bytes(); api._mhr = mhr
V = 4096
sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
allv = list(range(V))
sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
for k in range(400):
    sc.mix(1024, post_process=True)
    if k % 25 == 24: sc.sync()
sc.sync()
