# rocprofv3 kernel stats of the bench command per config, beside the bench line's own kernel time.
# gpurun -- "bash tools/r3_profile.sh [configs...]"   -> gpurun_out/r3p/ (copy the summaries into profiles/r3/)
export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O
for cfg in ${@:-3}; do
  rm -rf $O/prof_$cfg
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_$cfg -o cfg$cfg -- python $OLDPWD/bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $OLDPWD/$O/bench_config$cfg.json 2> $OLDPWD/$O/bench_config$cfg.err )
  f=$(find $O/prof_$cfg -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/config${cfg}_kernel_stats.csv
  python - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/config${cfg}_kernel_stats.csv")))
d = json.load(open("$O/bench_config$cfg.json"))
print("config $cfg: bench line: %.1f us/step, kernel %.2f us (HIP events bound to the dispatch), %s" % (d["ms_per_step"] * 1e3, d["roofline"]["kernel_ms"] * 1e3, d["roofline"].get("kernel", "")))
for r in rows[:8]:
    print("   %-90s calls %6s avg %8.2f us  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
