"""Worker of tests/test_multi_rank.py::test_overlapped_engine_single_rank_nccl (own process: torch
first, then liboalgpu.so)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "openal-soft_amd"))
import oracle_lib as ol          # noqa: E402
import oalgpu                    # noqa: E402
from oalgpu.shard import OverlappedGpuEngine, ShardedMixer   # noqa: E402


def main():
    assert oalgpu.device_count() > 0, "needs a HIP device"
    mhr = os.environ["OAL_TEST_MHR"]
    which = "ref" if ol.available("ref") else "port"
    L = ol.load(which)
    L.L.oal_set_simd(1)
    L.hrtf_load(mhr)
    api = oalgpu.Api(oalgpu.MATH_FAST)
    api.hrtf_load(mhr)
    rng = np.random.default_rng(8)
    cc = np.zeros((4, 128, 2), np.float32)
    cc[:, :64] = rng.uniform(-0.2, 0.2, (4, 64, 2))
    data = rng.uniform(-1, 1, 9000).astype(np.float32)
    nvoices = 40

    def params(v, k):
        r = np.random.default_rng(100 * v + k)
        return ol.make_voice_params(60211, ol.RS_BSINC24,
                                    hrtf=(np.arcsin(r.uniform(-1, 1)), r.uniform(-np.pi, np.pi), 2.0, 0.0, 0.05),
                                    direct_filter=ol.default_filter(active=v % 2, gain_hf=0.4))

    def build(lib, **kw):
        sc = lib.make_scene(num_dry=4, num_real=2, hrtf=True, **kw)
        sc.set_direct_hrtf(cc, [1.0, 0.8, 0.8, 0.8], 400.0 / 48000.0, 64)
        b = sc.add_buffer(data, ol.FMT_FLOAT, loop_start=0, loop_end=9000)
        for v in range(nvoices):
            sc.add_voice(b, looping=True, position=(v * 701) % 8000, frac=0)
            sc.set_params(v, params(v, 0))
        return sc

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        gsc, osc = build(api, max_voices=nvoices), build(L)
        engine = OverlappedGpuEngine(gsc, torch, 0)
        engine.always_reduce = True
        mixer = ShardedMixer(engine, dist, 0, 1)
        for k in range(5):
            for v in range(0, nvoices, 3):
                gsc.set_params(v, params(v, k + 1))
                osc.set_params(v, params(v, k + 1))
            mixer.update(1024)
            osc.mix(1024, post_process=True)
        got = np.concatenate([gsc.dry().ravel(), gsc.hrtf_accum().ravel()]).astype(np.float64)
        want = np.concatenate([osc.dry().ravel(), osc.hrtf_accum().ravel()]).astype(np.float64)
        err = np.abs(got - want).max()
        assert err <= 2e-5 * np.abs(want).max() + 1e-7, err
        gsc.close(); osc.close()
        print("overlapped ok, max err %.3e" % err)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
