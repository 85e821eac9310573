#!/bin/bash
# the resident launch's step, where its time goes (wait counters) and the kernels' trace, one gpurun call
set -u
tag=${1:-r5c}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout 300 python bench.py --resident on --no-cpu-baseline --steps 1000 --warmup 100 --repeats 3 > "$out/bench1000_on.log" 2>&1
timeout 300 python bench.py --resident off --no-cpu-baseline --steps 1000 --warmup 100 --repeats 3 > "$out/bench1000_off.log" 2>&1
timeout 300 python bench.py --resident on --no-cpu-baseline --steps 20 --warmup 5 > "$out/bench20_on.log" 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$out/prof_on" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --resident on --no-cpu-baseline --steps 200 --warmup 20 --repeats 1 --preroll 100 ) > "$out/prof_on.log" 2>&1
grep -h '"metric"' "$out"/bench*.log | python -c '
import sys, json
for line in sys.stdin:
    j = json.loads(line); r = j["roofline"]
    print("steps", j["steps"], "ms_per_step %.5f" % j["ms_per_step"], "kernel_ms %.5f" % r["kernel_ms"], "launched %.5f" % r.get("kernel_ms_launched", 0),
          "repeat", j["config"]["repeat_ms_per_step"]["median"], "resident", json.dumps(r.get("resident")))
' | tee "$out/summary.txt"
ls -la "$out/prof_on" | head
find "$out/prof_on" -name "*kernel_stats*" | head -1 | xargs -r head -12
