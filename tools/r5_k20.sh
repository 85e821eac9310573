#!/bin/bash
set -u
out=gpurun_out/${1:-r5k20}
mkdir -p "$out"
for v in "off" "off --xflags 16" "on"; do
  n=$(echo $v | tr -d ' -')
  for i in 1 2; do
  timeout 300 python bench.py --resident $v --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '"metric"' | python -c '
import sys, json
j = json.loads(sys.stdin.readline()); r = j["roofline"]
print("'"$n"' ms_per_step %.5f cold %.5f repeat %s kernel_ms %.5f" % (j["ms_per_step"], j["config"]["cold_block_ms_per_step"], j["config"]["repeat_ms_per_step"], r["kernel_ms"]))' | tee -a "$out/summary.txt"
  done
done
