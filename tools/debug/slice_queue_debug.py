"""which voice of tests/test_queue_adpcm.py's scene differs between the slice kernel and the stream-row kernel"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd"))
import oalgpu
import oracle_lib as ol
import test_queue_adpcm as T
from oalgpu import synth

mhr = "/tmp/synth_dbg.mhr"
synth.write_synth_mhr(mhr)
orig_make = None

def run_only(flags, only):
    api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=flags)
    real_make = api.make_scene
    holder = {}
    def make_scene(**kw):
        sc = real_make(**kw)
        holder["sc"] = sc
        real_mix = sc.mix
        first = {"done": False}
        def mix(n, post_process=False):
            if not first["done"]:
                first["done"] = True
                for u in range(16):
                    if u != only:
                        sc.set_state(u, ol.VOICE_STOPPED)
            return real_mix(n, post_process=post_process)
        sc.mix = mix
        return sc
    api.make_scene = make_scene
    out, ints = T.run(api, mhr, False, 2)
    return out, ints

for v in range(16):
    a, ia = run_only(128, v)
    b, ib = run_only(0, v)
    errs = [float(np.abs(x - y).max()) for x, y in zip(a, b)]
    scale = [float(np.abs(y).max()) for y in b]
    bad = [k for k in range(len(errs)) if errs[k] > 2e-5 * scale[k] + 1e-7]
    print("voice", v, "kind", v % 8, "step", [65536, 100000, 30000, 230000, 70001][v % 5], "errs", ["%.2e" % e for e in errs], "scale", ["%.2e" % s for s in scale], "BAD" if bad else "", "ints differ" if ia != ib else "")
    if bad:
        k = bad[0]
        d = np.abs(a[k] - b[k]); n = T.TODO[k]
        lines = d.reshape(-1, n) if d.size % n == 0 else None
        if lines is not None:
            for li in range(lines.shape[0]):
                if lines[li].max() > 1e-6:
                    idx = np.nonzero(lines[li] > 1e-6)[0]
                    print("   update", k, "line", li, "max", lines[li].max(), "frames", idx[:6], "...", idx[-3:], "count", idx.size)
