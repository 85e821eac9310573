"""Debug helper: run one SCENES entry on the GPU (fast/exact) and the oracle, print where they differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd"))
import numpy as np
import oracle_lib as ol, oalgpu
from oalgpu import synth
from scenes import SCENES, run_scene

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
mode = sys.argv[2] if len(sys.argv) > 2 else "fast"
nvo = int(sys.argv[3]) if len(sys.argv) > 3 else None
cfg = dict(SCENES[idx])
if nvo: cfg["nvoices"] = nvo
if len(sys.argv) > 4: cfg["n_updates"] = int(sys.argv[4])
if len(sys.argv) > 5: cfg["move"] = bool(int(sys.argv[5]))
L = ol.load("ref" if ol.available("ref") else "port"); L.L.oal_set_simd(1)
mhr = synth.write_synth_mhr("/tmp/synth_dbg.mhr")
api = oalgpu.Api(oalgpu.MATH_FAST if mode == "fast" else oalgpu.MATH_EXACT)
fa, ia = run_scene(api, mhr, rng_seed=idx + 1, **cfg)
fb, ib = run_scene(L, mhr, rng_seed=idx + 1, **cfg)
print("ints equal:", ia == ib)
if ia != ib:
    for v, (a, b) in enumerate(zip(ia, ib)):
        if a != b: print("  voice", v, a, b)
hrtf = cfg["hrtf"]; sends = cfg.get("sends", 0)
labels = []
ndry = 6 if hrtf else 5
for k in range(cfg["n_updates"]):
    labels.append((f"u{k}.dry", ndry * 1024))
    if hrtf: labels.append((f"u{k}.accum", 1152 * 2))
    for sl in range(2 if sends else 0): labels.append((f"u{k}.wet{sl}", 4 * 1024))
for v in range(cfg["nvoices"]):
    labels += [(f"v{v}.prev", 48), (f"v{v}.drycur", 32), (f"v{v}.hist", 64), (f"v{v}.misc", 25)]
    for i in range(sends): labels += [(f"v{v}.sendcur{i}", 25), (f"v{v}.sendlp{i}", 12)]
pos = 0
scale = np.max(np.abs(fb))
for name, n in labels:
    a, b = fa[pos:pos + n].astype(np.float64), fb[pos:pos + n].astype(np.float64)
    err = np.max(np.abs(a - b)) if n else 0
    if err > 2e-5 * scale:
        w = int(np.argmax(np.abs(a - b)))
        bad = np.flatnonzero(np.abs(a - b) > 2e-5 * scale)
        print('   bad idx', bad[:12], '...', bad[-6:])
        print(f"{name:14s} n={n:5d} maxerr {err:.3e} at {w} (gpu {a[w]:.6f} ref {b[w]:.6f}) max|ref| {np.max(np.abs(b)):.3e} nbad {int(np.sum(np.abs(a-b) > 2e-5*scale))}")
    pos += n
assert pos == fa.size == fb.size, (pos, fa.size, fb.size)
print("done; scale", scale)
