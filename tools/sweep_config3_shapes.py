import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import tempfile
import oalgpu
from oalgpu import synth
import bench
import test_gpu_baseline_configs as t
f = tempfile.NamedTemporaryFile(suffix=".mhr", delete=False); f.write(synth.synth_mhr_bytes()); f.close()
orig = bench.build_scene
origApi = oalgpu.Api
for flags in (0,):
    oalgpu.Api = lambda mode, flags=flags: origApi(mode, ctx_flags=flags)
    for vpg in (8,):
        bench.build_scene = lambda a, b, c, d, e, g, h, i, vpg=vpg: orig(a, b, c, d, e, g, h, vpg)
        for V in (3072, 4096, 3072, 4096):
            try:
                t.run_config(3, V, f.name, todo=(1024, 1024))
                print("flags", flags, "vpg", vpg, "V", V, "ok")
            except AssertionError as e:
                print("flags", flags, "vpg", vpg, "V", V, "FAIL", str(e)[:200])
