"""Voice sharding over ranks on CPU: two gloo processes run the
orchestration bench.py uses on GPUs (oalgpu.shard.ShardedMixer: partial buses per rank, ONE
sum-reduce of the bus block, effects + HRTF post-process on rank 0 which alone carries the
accumulator tail) with the CPU oracle as each rank's mixer, and rank 0 checks the result against
the same scene mixed unsharded.  Also the shard arithmetic itself."""
import os
import socket
import subprocess
import sys

import pytest

import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "openal-soft_amd"))


def test_shard_range_partitions_voices():
    from oalgpu.shard import shard_range
    for total in (0, 1, 7, 22, 4096, 32768):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_weighted_shards_balance_cost_classes():
    from oalgpu.shard import voice_cost, weighted_shards
    # BASELINE config 4 mix: HRTF-free voices with v % 5 sends, every fourth filtered
    costs = [voice_cost(False, 24, v % 5, v % 4 == 1) for v in range(8192)]
    for world in (1, 2, 3, 8):
        shards = weighted_shards(costs, world)
        assert sorted(v for s in shards for v in s) == list(range(8192))
        loads = [sum(costs[v] for v in s) for s in shards]
        assert max(loads) - min(loads) <= max(costs) + 1e-9
    # work only rank 0 has (effects, post-process) is taken off its share
    extra = 0.1 * sum(costs)
    shards = weighted_shards(costs, 8, rank0_extra=extra)
    loads = [sum(costs[v] for v in s) for s in shards]
    assert loads[0] + extra <= max(loads[1:]) + max(costs) and loads[0] < min(loads[1:])
    # an all-HRTF scene next to a dry one: HRTF voices weigh ~3x
    assert 2.0 < voice_cost(True) / voice_cost(False) < 4.0


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_scene_matches_unsharded(synth_mhr, world):
    if not ol.available("port"):
        pytest.skip("oracle/liboalport.so not built")
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OAL_TEST_MHR=synth_mhr, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "multi_rank_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
    assert "update 3" in outs[0]


@pytest.mark.gpu
def test_library_comm_single_rank_rccl(synth_mhr):
    """The GPU side of the sharded update as the library runs it: oalgpu_comm_init over a ONE-rank
    RCCL communicator, then oalgpu_mix_update (voice kernel on the main stream; partial-bus reduction,
    ncclReduce of the bus block and the post-process on the post stream) against the oracle, over several
    back-to-back updates without draining in between.  In its own process (librccl.so is loaded there)."""
    # (once in some forty runs on the pool's boxes the worker did not finish within 300 s -- where it hung was not seen; the same
    # tree ran it three times in a row in seconds on the next box: one retry, on another port, with a shorter limit)
    p = None
    for attempt in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), OAL_TEST_MHR=synth_mhr)
        try:
            p = subprocess.run([sys.executable, os.path.join(HERE, "overlapped_worker.py")], env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=120)
            break
        except subprocess.TimeoutExpired:
            if attempt == 1:
                raise
    assert p.returncode == 0, p.stdout[-3000:]
    assert "overlapped ok" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("config", [3, 4, 5, "3 with parameter blocks installed by the voice kernels"])
def test_library_sharded_update_two_processes_one_gpu(synth_mhr, tmp_path, config):
    """The library's N > 1 path itself: two processes share the GPU, each with a context of its own, connected by
    oalgpu_comm_init_host (RCCL refuses two ranks on one device; the host-staged transport sits behind the same
    interface as the ncclReduce).  Rank 1 mixes its shard without effects, post-process or accumulator carry and
    hands its bus block over; rank 0 sums, runs the slots' effects and the post-process.  Ten pipelined updates
    (the first eight without a host synchronisation: twice the depth of the transport's ring), voices dealt by cost
    class; rank 0's buses and every voice's integer state must equal the scene mixed unsharded.  Config 3 (HRTF),
    config 4 (dry lines + sends into four EAX reverb slots) and config 5 -- BASELINE configs[4]'s sharded shape: rank 1's
    send rows reach the wet bus of rank 0's 65 536-tap convolution slot through the reduce
    (alc/effects/convolution.cpp:623-716 runs where the summed wet bus is)."""
    import numpy as np
    env = dict(os.environ)
    if not isinstance(config, int):             # (what bench.py --gpus N runs per rank: a parameter block per update, OALGPU_CTX_APPLY_IN_VOICE_KERNEL)
        config = 3
        env["OALGPU_TEST_PARAM_BLOCKS"] = "1"
    total = 600
    name = f"/oalgpu_test_{os.getpid()}_{config}_{len(env) % 7}"
    prefix = str(tmp_path / f"c{config}")
    procs = []
    for rank in range(2):
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "host_transport_worker.py"), str(config), str(rank), "2",
                                       name, str(total), prefix, synth_mhr], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out[-3000:]}"
    r0 = np.load(prefix + "_rank0.npz")
    r1 = np.load(prefix + "_rank1.npz")
    assert sorted(list(r0["voices"]) + list(r1["voices"])) == list(range(total)) and len(r1["voices"]) > total // 4
    checked = 0
    for key in r0.files:
        if key.startswith("whole_") and key != "whole_ints":
            got, want = r0[key[len("whole_"):]].astype(np.float64), r0[key].astype(np.float64)
            scale = np.abs(want).max()
            err = np.abs(got - want).max()
            assert err <= 2e-5 * scale + 1e-7, (config, key, err, scale)
            checked += scale > 1e-4
    assert checked >= 2, "the compared buses must carry sound"
    whole = {int(r[0]): tuple(int(x) for x in r[1:]) for r in r0["whole_ints"]}
    for r in list(r0["ints"]) + list(r1["ints"]):
        assert whole[int(r[0])] == tuple(int(x) for x in r[1:]), (config, r)
