"""Debug aid: which voices differ between the two FIR forms after one update (filter state, history)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import oalgpu
from oalgpu import synth
import bench

def run(flags, V, vpg):
    api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=flags)
    mhr = synth.synth_mhr_bytes(); api._mhr = mhr
    sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, vpg)
    allv = list(range(V))
    sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
    sc.mix(1024, post_process=False)
    acc = sc.hrtf_accum().copy()
    st = []
    for v in range(V):
        s = sc.voice_state(v)
        st.append((np.array(list(s.hrtf_history), np.float32), s.direct_lp.z1, s.direct_lp.z2, s.direct_hp.z1, s.direct_hp.z2,
                   [getattr(s.direct_hp, k) for k in ("b0", "b1", "b2", "a1", "a2")], [getattr(s.direct_lp, k) for k in ("b0", "b1", "b2", "a1", "a2")]))
    sc.close()
    return acc, st, script

V, vpg = 4096, 8
a, sa, script = run(0, V, vpg)
a2, sa2, _ = run(0, V, vpg)
print("mf run 1 vs run 2: voices differing", [v for v in range(V) if abs(sa[v][3] - sa2[v][3]) > 1e-5])
b, sb, _ = run(oalgpu.CTX_FIR_VALU, V, vpg)
print("accum max diff", np.abs(a - b).max())
bad = []
for v in range(V):
    dh = np.abs(sa[v][0] - sb[v][0]).max(); dz = max(abs(sa[v][i] - sb[v][i]) for i in range(1, 5))
    if dh > 1e-5 or dz > 1e-5:
        bad.append(v)
        if len(bad) <= 24:
            print("voice", v, "wg", v // vpg, "slot", v % vpg, "gv", script.gv(v), "filtered", script.filter_active(v), "moving", script.is_moving(v),
                  "hist diff %.3e" % dh, "z diff %.3e" % dz, "z mf", sa[v][1:5], "z valu", sb[v][1:5], "hp", sa[v][5], sb[v][5] == sa[v][5], "lp", sa[v][6])
print(len(bad), "voices differ; slots:", np.bincount([v % vpg for v in bad], minlength=vpg), "wg%8:", np.bincount([(v // vpg) % 8 for v in bad], minlength=8))
print("wgs:", sorted(set(v // vpg for v in bad))[:80])
