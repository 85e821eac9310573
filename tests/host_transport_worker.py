"""Worker of tests/test_multi_rank.py::test_library_sharded_update_two_processes_one_gpu: one rank of a
voice-sharded scene through the LIBRARY's own N > 1 code -- oalgpu_comm_init_host (the host-staged transport:
several processes may share one GPU, which RCCL refuses), then pipelined oalgpu_mix_update calls: rank > 0 mixes
its shard and hands its bus block over, rank 0 sums, runs the effect slots and the post-process and alone
carries the HRTF accumulator.  Voices are dealt by cost class (oalgpu.shard.weighted_shards).  Rank 0 then
mixes the whole scene on a context without a communicator; everything is written to <out>.npz for the test
to compare.   argv: config rank world shm_name total_voices out_prefix mhr_path"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd"))
sys.path.insert(0, ROOT)
import oalgpu                                  # noqa: E402
from oalgpu import synth                       # noqa: E402
from oalgpu.shard import voice_cost, weighted_shards   # noqa: E402
import bench                                   # noqa: E402

SIZES = (1024, 1024, 1024, 1024, 700, 1024, 1024, 1024, 1024, 1024)
READ_AFTER = (7, 9)                            # the reads drain the pipeline: eight updates back to back first -- twice the
                                               # depth of the transport's ring, so the ranks do throttle each other


BLOCKS = os.environ.get("OALGPU_TEST_PARAM_BLOCKS") == "1"
held = []                                       # (a block stays alive until the update that installs it has run)


def run(sc, script, nslots, hrtf):
    voices = list(range(script.nvoices))
    moving = [v for v in voices if script.is_moving(v)]
    out = {}
    for k, n in enumerate(SIZES):
        vs = voices if k == 0 else moving
        if vs and BLOCKS and k:
            # bench.py's N > 1 form: a parameter block per update, installed by the rank's own voice kernel (OALGPU_CTX_APPLY_IN_VOICE_KERNEL)
            blk = sc.param_block(vs, bench.param_array(oalgpu, script, vs, k))
            sc.apply_block(blk)
            held.append(blk)
        elif vs:
            sc.set_params_batch(vs, bench.param_array(oalgpu, script, vs, k))
        sc.mix(n, post_process=True)
        if k in READ_AFTER:
            out[f"dry{k}"] = sc.dry().copy()
            for s in range(nslots):
                out[f"wet{k}_{s}"] = sc.wet(s).copy()
            if hrtf:
                out[f"acc{k}"] = sc.hrtf_accum().copy()
    sc.sync()
    ints = []
    for v in voices:
        st = sc.voice_state(v)
        ints.append((script.gv(v), st.play_state, st.position, st.position_frac, st.fading))
    out["ints"] = np.array(ints, np.int64)
    return out


def main():
    config, rank, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    name, total, prefix, mhr_path = sys.argv[4], int(sys.argv[5]), sys.argv[6], sys.argv[7]
    hrtf = config in (3, 5)
    nslots = {4: 4, 5: 1}.get(config, 0)
    with open(mhr_path, "rb") as f:
        mhr = f.read()
    api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=oalgpu.CTX_APPLY_IN_VOICE_KERNEL if BLOCKS else 0)
    api._mhr = mhr
    probe = synth.SceneScript(config, total)
    costs = [voice_cost(hrtf, 24, {4: v % 5, 5: 1}.get(config, 0), probe.filter_active(v)) for v in range(total)]
    shards = weighted_shards(costs, world, rank0_extra=0.05 * sum(costs) / world)
    mine = shards[rank]
    sc, script = bench.build_scene(oalgpu, synth, api, config, len(mine), 0, mhr, 0, voice_map=mine)
    sc.comm_init_host(name, rank, world)
    res = run(sc, script, nslots if rank == 0 else 0, hrtf and rank == 0)
    res["voices"] = np.array(mine, np.int64)
    sc.comm_destroy()
    sc.close()
    if rank == 0:
        whole, wscript = bench.build_scene(oalgpu, synth, api, config, total, 0, mhr, 0)
        ref = run(whole, wscript, nslots, hrtf)
        whole.close()
        for k, v in ref.items():
            res["whole_" + k] = v
    np.savez(prefix + f"_rank{rank}.npz", **res)
    print(f"rank {rank} of {world}: {len(mine)} voices done", flush=True)


if __name__ == "__main__":
    main()
