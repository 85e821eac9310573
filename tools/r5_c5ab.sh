#!/bin/bash
# config 5 (rank 0 of every N > 1 run): register line accumulators (240 VGPRs: nothing fits beside) against stream rows
# (223: the reduction and the post-process co-reside as in config 3), interleaved
set -u
out=gpurun_out/${1:-r5c5ab}
mkdir -p "$out"
for rep in 1 2 3; do
  for x in 0 8; do
    timeout 300 python bench.py --config 5 --steps 300 --warmup 20 --repeats 3 --no-cpu-baseline --xflags $x > "$out/c5_x${x}_$rep.json" 2> "$out/c5_x${x}_$rep.err"
    python - "$out/c5_x${x}_$rep.json" $x <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("xflags", sys.argv[2], "ms_per_step %.5f kernel_ms %.5f  %s" % (j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["kernel"]))
PY
  done
done
