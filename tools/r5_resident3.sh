#!/bin/bash
set -u
tag=${1:-r5d}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_resident.py tests/test_gpu_pipeline.py tests/test_gpu_async_boundary.py -q > "$out/tests.log" 2>&1; tail -2 "$out/tests.log"
for v in "on" "off" "on --static" "off --static"; do
  n=$(echo $v | tr -d ' -')
  timeout 300 python bench.py --resident $v --no-cpu-baseline --steps 1000 --warmup 100 --repeats 3 > "$out/bench1000_$n.log" 2>&1
done
grep -H '"metric"' "$out"/bench*.log | python -c '
import sys, json
for line in sys.stdin:
    name, rest = line.split(":", 1)
    j = json.loads(rest); r = j["roofline"]
    res = r.get("resident") or {}
    print(name.split("/")[-1], "ms_per_step %.5f" % j["ms_per_step"], "kernel_ms %.5f" % r["kernel_ms"], "launched %.5f" % r.get("kernel_ms_launched", 0),
          "repeat %.5f" % j["config"]["repeat_ms_per_step"]["median"], json.dumps(res.get("waits_us_per_update")))
' | tee "$out/summary.txt"
