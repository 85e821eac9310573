"""What a K-update block costs with the voice kernel resident (OALGPU_CTX_RESIDENT) and launched per update: blocks of L updates
bracketed by oalgpu_sync on both sides (the bench contract's shape), several L, median of R blocks each; the intercept of the
line through (L, time) is what a block pays once -- the launch, its workgroups' cold start, the drain of the last update's
reduction and post-process -- the slope the steady period.  usage: python tools/resident_block_cost.py [--voices 4096]"""
import argparse
import gc
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voices", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=15)
    args = ap.parse_args()
    import oalgpu
    from oalgpu import synth
    import bench
    mhr = open(os.path.join(ROOT, "tests", "golden", "default_hrtf.mhr"), "rb").read()
    lengths = (1, 2, 5, 10, 20, 50, 100, 200)
    for mode in ("resident", "launched"):
        api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=oalgpu.CTX_RESIDENT if mode == "resident" else 0)
        api._mhr = mhr
        V = args.voices
        sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
        allv = list(range(V))
        moving = [v for v in allv if script.is_moving(v)]
        sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
        blocks = [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(48)]
        gc.collect(); gc.disable()
        k = 0
        for _ in range(600):                      # clocks up, queues warm
            sc.apply_block(blocks[k % 48]); sc.mix(1024, post_process=True); k += 1
            if k % 25 == 0:
                sc.sync()
        sc.sync()
        rows = []
        prev = sc.resident_stats() if mode == "resident" else None
        for L in lengths:
            t = []
            for _ in range(args.reps):
                sc.sync()
                t0 = time.perf_counter()
                for _ in range(L):
                    sc.apply_block(blocks[k % 48]); sc.mix(1024, post_process=True); k += 1
                sc.sync()
                t.append((time.perf_counter() - t0) * 1e6)
            rows.append((L, float(np.median(t)), float(np.min(t))))
            if mode == "resident":
                st = sc.resident_stats()
                if L in (20, 200) and prev is not None:
                    n = st["updates"] - prev["updates"]
                    print(f"   L = {L}: per update [us] " + ", ".join(f"{k[:-3]} {(st[k] - prev[k]) / n:.2f}" for k in st if k.endswith("_us")))
                prev = st
        A = np.array([[L, 1.0] for L, _, _ in rows[2:]])
        slope, icpt = np.linalg.lstsq(A, np.array([m for _, m, _ in rows[2:]]), rcond=None)[0]
        print(f"{mode}: per update {slope:.2f} us, per block {icpt:.1f} us")
        for L, m, mn in rows:
            print(f"   L = {L:4d}: median {m:9.1f} us ({m / L:7.2f} per update), min {mn:9.1f}")
        if mode == "resident":
            print("   ", sc.resident_stats())
        gc.enable()
        sc.close()


if __name__ == "__main__":
    main()
