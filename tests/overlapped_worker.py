"""Worker of tests/test_multi_rank.py::test_library_comm_single_rank_rccl (own process): the
library's multi-GPU path -- oalgpu_comm_init, then oalgpu_mix_update with the ncclReduce of the bus
block issued by the library on its post stream -- over a ONE-rank RCCL communicator, against the
oracle."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "openal-soft_amd"))
import oracle_lib as ol          # noqa: E402
import oalgpu                    # noqa: E402


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

    if True:
        gsc, osc = build(api, max_voices=nvoices), build(L)
        gsc.comm_init(oalgpu.comm_unique_id(), 0, 1)
        for k in range(5):
            for v in range(0, nvoices, 3):
                gsc.set_params(v, params(v, k + 1))
                osc.set_params(v, params(v, k + 1))
            gsc.mix(1024, post_process=True)         # voices | reduction + ncclReduce + post-process, pipelined
            osc.mix(1024, post_process=True)
        got = np.concatenate([gsc.dry().ravel(), gsc.hrtf_accum().ravel()]).astype(np.float64)
        want = np.concatenate([osc.dry().ravel(), osc.hrtf_accum().ravel()]).astype(np.float64)
        err = np.abs(got - want).max()
        assert err <= 2e-5 * np.abs(want).max() + 1e-7, err
        gsc.comm_destroy()
        gsc.close(); osc.close()
        # a dry-line context (no HRTF: the one-stream path, ncclReduce on the main stream behind the bus
        # reduction), one send into a slot
        def build_dry(lib, **kw):
            sc = lib.make_scene(num_dry=5, num_real=0, num_sends=1, num_slots=1, wet_channels=4, hrtf=False, **kw)
            b = sc.add_buffer(data, ol.FMT_FLOAT, loop_start=0, loop_end=9000)
            r = np.random.default_rng(3)
            for v in range(nvoices):
                sc.add_voice(b, looping=True, position=(v * 701) % 8000, frac=0)
                sc.set_params(v, ol.make_voice_params(60211, ol.RS_BSINC24, dry_gains=r.uniform(0, 0.1, 5),
                                                      direct_filter=ol.default_filter(active=v % 2, gain_hf=0.4),
                                                      sends=[(0, r.uniform(0.05, 0.3, 4), None)]))
            return sc
        gsc, osc = build_dry(api, max_voices=nvoices), build_dry(L)
        gsc.comm_init(oalgpu.comm_unique_id(), 0, 1)
        for k in range(3):
            gsc.mix(1024, post_process=True)
            osc.mix(1024, post_process=False)
        got = np.concatenate([gsc.dry().ravel(), gsc.wet(0).ravel()]).astype(np.float64)
        want = np.concatenate([osc.dry().ravel(), osc.wet(0).ravel()]).astype(np.float64)
        err2 = np.abs(got - want).max()
        assert err2 <= 2e-5 * np.abs(want).max() + 1e-7, err2
        gsc.close(); osc.close()                      # (the context destroys its communicator)
        print("overlapped ok, max err %.3e / %.3e" % (err, err2))


if __name__ == "__main__":
    main()
