"""Debug aid: the HRTF voice kernel's two FIR forms against each other on the bench scene (GPU vs GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import oalgpu
from oalgpu import synth
import bench

def run(flags, V, vpg, updates=2):
    api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=flags)
    mhr = synth.synth_mhr_bytes(); api._mhr = mhr
    sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, vpg)
    allv = list(range(V)); moving = [v for v in allv if script.is_moving(v)]
    outs = []
    for k in range(updates):
        voices = allv if k == 0 else moving
        sc.set_params_batch(voices, bench.param_array(oalgpu, script, voices, k))
        sc.mix(1024, post_process=False)
        outs.append(sc.hrtf_accum().copy())
    name = sc.voice_kernel_name()
    sc.close()
    return outs, name

for V, vpg in ((4, 4), (8, 8), (16, 8), (16, 4), (64, 8), (4096, 0)):
    a, na = run(0, V, vpg)
    b, nb = run(oalgpu.CTX_FIR_VALU, V, vpg)
    for k in range(len(a)):
        d = np.abs(a[k] - b[k]); m = np.abs(b[k]).max()
        fr = np.argmax(d.max(axis=-1) if d.ndim > 1 else d)
        print(f"V={V} vpg={vpg} update {k}: {na} vs {nb}: max diff {d.max():.3e} of {m:.3e} at flat index {int(np.argmax(d))} shape {d.shape}")
        dd = d.reshape(-1, 2) if d.size % 2 == 0 else d
        per64 = dd.max(axis=1)[:1088].reshape(17, 64).max(axis=1)
        print("   per 64-frame block:", " ".join(f"{x:.1e}" for x in per64))
