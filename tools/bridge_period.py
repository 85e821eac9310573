"""What the reference's own loop gets from the binding that is shipped (include/oalgpu_openal.hpp, compiled against the reference
by oracle/ref_bridge.cpp): N HRTF sources (BASELINE configs[2]: bsinc24, Default HRTF.mhr, a quarter filtered) rendered through
DeviceBase::renderSamples -- ProcessContexts, CalcVoiceParams for the sources that moved, the voice loop, the reference's own
HRTF post-process -- with Voice::mix (a) the reference's own (one mixer thread: its real operating mode), (b) the batch mixer
comparing every voice's parameters, (c) the batch mixer told which voices CalcSourceParams recomputed (its hook).  Per update,
median of the timed updates; "moving": every 4th source gets a new direction per update, "static": none does.
usage: python tools/bridge_period.py [--sources 4096] [--updates 40]   (needs oracle/_ref/liboalbridge.so and a GPU)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sources", type=int, default=4096)
    ap.add_argument("--updates", type=int, default=40)
    args = ap.parse_args()
    import bridge_lib as bl
    import oracle_lib as ol
    import oalgpu
    N = args.sources
    print(f"{N} HRTF sources behind the reference's renderSamples, 1024-sample updates, one host thread")
    print(f"{'':34s}{'us / update':>12s}{'voices/s':>12s}{'of which outside render: moves':>34s}")
    for scene in ("moving", "static"):
        for name, mode, track, depth, hook in (("reference Voice::mix (CPU)", bl.MODE_CPU, False, 0, False), ("batch mixer, every voice compared", bl.MODE_BATCH, False, 0, False),
                                         ("batch mixer + parameter hook", bl.MODE_BATCH, True, 0, False),
                                         ("batch mixer pipelined (post-process on the GPU, output 2 updates late) + hook", bl.MODE_BATCH, True, 2, False),
                                         ("batch mixer pipelined + the hooks inside alc/alu.cpp (directions, not responses)", bl.MODE_BATCH, False, 2, True)):
            b = bl.Bridge(mode, oalgpu.MATH_FAST, hrtf=True, num_sends=0)
            if depth:
                b.set_pipelined(depth)
            if track:
                b.track_changes(True)
            if hook:
                b.hook_alu(True)
            srcs = bl.build_config3(b, N, slot=-1)
            for k in range(12):                                 # voices started, the sources that run out of buffer gone, clocks up
                b.render(1024)
            t_render, t_move = [], []
            bt0 = b.batch_times() if depth else None
            for k in range(args.updates):
                t0 = time.perf_counter()
                if scene == "moving":
                    bl.move_config3(b, srcs, k + 1, slot=-1)    # (the application's side: new VoiceProps for a quarter of the sources)
                t1 = time.perf_counter()
                b.render(1024)
                t2 = time.perf_counter()
                t_move.append(t1 - t0); t_render.append(t2 - t1)
            parts = ""
            if depth:
                bt1 = b.batch_times()
                w, sub, col = [(a1 - a0) / args.updates * 1e6 for a0, a1 in zip(bt0, bt1)]
                parts = f"   [mean per update: flush walks the voices {w:.1f} us, submits {sub:.1f}, collects {col:.1f}; the rest is the reference's ProcessContexts and the per-voice seam]"
            if depth:
                parts += "\n      render times of the timed updates [us]: " + " ".join(f"{t * 1e6:.0f}" for t in t_render)
            b.close()
            r = float(np.median(t_render))
            print(f"{scene + ': ' + name:34s}{r * 1e6:12.1f}{N / r / 1e6:11.2f}M{float(np.median(t_move)) * 1e6:34.1f}{parts}", flush=True)


if __name__ == "__main__":
    main()
