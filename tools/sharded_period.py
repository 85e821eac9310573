"""Profiling aid: period of back-to-back updates with the library's multi-GPU path switched on over a
ONE-rank RCCL communicator (oalgpu_comm_init: the ncclReduce of the bus block rides on the post stream)
against the plain pipelined update.  gpurun -- "python tools/sharded_period.py" """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oalgpu
from oalgpu import synth
import bench
V = 4096
api = oalgpu.Api(oalgpu.MATH_FAST)
mhr = synth.synth_mhr_bytes(); api._mhr = mhr
for comm in (False, True):
    sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
    allv = list(range(V)); moving = [v for v in allv if script.is_moving(v)]
    sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
    blocks = [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(32)]
    if comm:
        sc.comm_init(oalgpu.comm_unique_id(), 0, 1)
    for k in range(300):
        sc.apply_block(blocks[k % 32]); sc.mix(1024, post_process=True)
    sc.sync()
    t0 = time.perf_counter()
    n = 1000
    for k in range(n):
        sc.apply_block(blocks[k % 32]); sc.mix(1024, post_process=True)
    sc.sync()
    print("ncclReduce in the update" if comm else "plain pipelined update  ", "period %.2f us" % ((time.perf_counter() - t0) / n * 1e6))
    sc.close()
