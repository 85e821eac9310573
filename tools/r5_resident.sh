#!/bin/bash
# Round 5, the resident voice kernel on the GPU: parity (bit equality against one launch per update), then the A/B of the
# bench step with the mode on and off, in ONE gpurun call (boxes differ by up to 15 %: only numbers of one call are compared).
# usage (from the repo root on the GPU box): tools/r5_resident.sh [tag]
set -u
tag=${1:-r5a}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout "${TMO:-420}" "$@" ) > "$out/$name.log" 2>&1; echo "rc=$?" >> "$out/$name.log"; tail -3 "$out/$name.log"; }
run test_resident python -m pytest tests/test_gpu_resident.py -x -q
run test_pipeline python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_async_boundary.py -x -q
for mode in off on; do
    TMO=300 run bench20_$mode python bench.py --resident $mode --no-cpu-baseline --steps 20 --warmup 5
    TMO=300 run bench1000_$mode python bench.py --resident $mode --no-cpu-baseline --steps 1000 --warmup 100 --repeats 3
done
grep -h '"metric"' "$out"/bench*.log | python -c '
import sys, json
for line in sys.stdin:
    j = json.loads(line)
    r = j["roofline"]
    print("steps", j["steps"], "ms_per_step %.5f" % j["ms_per_step"], "value %.1f M" % (j["value"] / 1e6), "kernel_ms %.5f" % r["kernel_ms"],
          "launched %.5f" % r.get("kernel_ms_launched", 0), "resident", (r.get("resident") or {}).get("updates_per_launch"),
          "repeat", j["config"]["repeat_ms_per_step"], "e2e", (j["config"].get("e2e_throughput") or {}).get("ms_per_update"))
' | tee "$out/summary.txt"
