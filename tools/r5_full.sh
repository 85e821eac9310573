#!/bin/bash
# the whole GPU suite, then the bench line as the driver asks for it (K = 20) and with the defaults (K = 1000)
set -u
out=gpurun_out/${1:-r5full}
mkdir -p "$out"
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > "$out/pytest_gpu.log" 2>&1; tail -4 "$out/pytest_gpu.log"
timeout 600 python bench.py --steps 20 --warmup 5 > "$out/bench_k20.json" 2> "$out/bench_k20.err"; tail -c 600 "$out/bench_k20.json" | head -c 300; echo
timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
python - "$out" <<'PY'
import json, sys
for n in ("bench_k20", "bench_default"):
    try:
        j = json.loads(open(f"{sys.argv[1]}/{n}.json").read().strip().splitlines()[-1])
        r = j["roofline"]
        print(n, "ms_per_step %.5f value %.1fM kernel_ms %.5f frac %.3f mode: %s" % (j["ms_per_step"], j["value"] / 1e6, r["kernel_ms"], r["frac"], j["config"]["voice_kernel_mode"]))
        print("   repeat", j["config"]["repeat_ms_per_step"], "cold", j["config"]["cold_block_ms_per_step"], "cpu", (j.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(n, "failed:", e)
PY
