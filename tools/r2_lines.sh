# dry-line / send contexts (BASELINE configs 2, 4, 5) after a change to the stream-row path:
# parity tests that cover them, then bench + kernel stats per config.  gpurun -- "bash tools/r2_lines.sh"
export TMPDIR=/tmp
O=gpurun_out/lines
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "not baseline_configs" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -x 2>&1 | tail -5
for c in 2 4 5; do
  timeout 300 python bench.py --config $c --steps 300 --warmup 20 --no-cpu-baseline < /dev/null > $O/bench_config$c.json 2>$O/bench_config$c.err
  python - <<PY
import json
d=json.load(open("$O/bench_config$c.json"))
print("config $c: %.1f us/step" % (d["ms_per_step"]*1000), d["roofline"]["kernel"], "%.1f us" % (d["roofline"]["kernel_ms"]*1000), d["config"].get("repeat_ms_per_step"))
PY
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c$c -o p -- python bench.py --config $c --steps 100 --warmup 5 --repeats 0 --no-cpu-baseline < /dev/null > $O/prof_c$c.log 2>&1
  cp $O/prof_c$c/p_kernel_stats.csv $O/config${c}_kernel_stats.csv
  head -7 $O/config${c}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
done
