"""Host time and period of the N>1 update path on one GPU (a 1-rank RCCL group): voices -> partial-bus
reduction -> RCCL reduce to rank 0 -> post-process, as bench.py --gpus N runs it per rank."""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, ROOT)
import torch, torch.distributed as dist, oalgpu
from oalgpu import synth
from oalgpu.shard import OverlappedGpuEngine, ShardedMixer
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
api = oalgpu.Api(oalgpu.MATH_FAST, device=0)
mhr = synth.synth_mhr_bytes(); api._mhr = mhr
V = 4096
sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
allv = list(range(V)); moving = [v for v in allv if script.is_moving(v)]
sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
blocks = [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(40)]
engine = OverlappedGpuEngine(sc, torch, 0); engine.always_reduce = True
mixer = ShardedMixer(engine, dist, 0, 1)
def run(name, fn, N=400):
    for k in range(30): fn(k)
    sc.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(N): fn(k)
    t1 = time.perf_counter(); sc.sync(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-44s host %.1f us  period %.1f us" % (name, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
run("apply + sharded update (torch reduce)", lambda k: (sc.apply_block(blocks[k % 40]), mixer.update(1024)))
engine.always_reduce = False
run("apply + sharded update without the reduce", lambda k: (sc.apply_block(blocks[k % 40]), mixer.update(1024)))
bus = engine.bus_tensor()
def only_reduce(k):
    with engine.collective():
        dist.reduce(bus, dst=0, op=dist.ReduceOp.SUM)
run("the reduce alone", only_reduce)
dist.destroy_process_group()
