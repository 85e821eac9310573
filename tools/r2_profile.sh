# Round-2 evidence for profiles/r2: the bench line, rocprofv3 kernel stats per BASELINE config, HBM PMC passes
# (FETCH_SIZE / WRITE_SIZE in separate passes) for configs 2-5.  gpurun -- "bash tools/r2_profile.sh"
export TMPDIR=/tmp
O=gpurun_out/r2
mkdir -p $O
timeout 900 python bench.py < /dev/null > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json
lscpu | head -20 > $O/gpu_box_lscpu.txt
for c in 3 2 4 5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c$c -o p -- python bench.py --config $c --steps 100 --warmup 5 --repeats 0 --no-cpu-baseline < /dev/null > $O/prof_c$c.log 2>&1
  cp $O/prof_c$c/p_kernel_stats.csv $O/config${c}_kernel_stats.csv
  head -6 $O/config${c}_kernel_stats.csv | cut -c1-150
  timeout 300 python bench.py --config $c --steps 300 --warmup 20 --no-cpu-baseline < /dev/null > $O/bench_config$c.json 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_c${c}_$ctr -o pmc -- python bench.py --config $c --steps 20 --warmup 3 --repeats 0 --no-cpu-baseline < /dev/null > $O/pmc_c${c}_$ctr.log 2>&1
  done
done
python - <<'PY'
import csv, json, collections
out = {}
for c in (2, 3, 4, 5):
    per = collections.defaultdict(dict)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            rows = [r for r in csv.DictReader(open(f"gpurun_out/r2/pmc_c{c}_{ctr}/pmc_counter_collection.csv")) if r["Counter_Name"] == ctr]
        except OSError:
            continue
        byk = collections.defaultdict(list)
        for r in rows: byk[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in byk.items():
            v.sort(); per[k][ctr] = {"median_kb": v[len(v) // 2], "calls": len(v)}
    out[c] = per
json.dump(out, open("gpurun_out/r2/pmc_hbm_by_kernel.json", "w"), indent=1)
for c, per in out.items():
    for k, d in per.items():
        if any(x in k for x in ("VoiceWave", "VoiceBlock", "LinesMix", "Conv", "Reverb")):
            print(c, k[:70], {n: round(x["median_kb"]) for n, x in d.items()})
PY
# the convolution kernel alone, the voice kernel's phases and its instruction-cache counters
python tools/conv_period.py > $O/conv_period.txt 2>&1; python tools/conv_period.py 8192 >> $O/conv_period.txt 2>&1; cat $O/conv_period.txt
timeout 300 python tools/phase_times.py > $O/voice_kernel_phase_times.txt 2>&1; tail -3 $O/voice_kernel_phase_times.txt
bash tools/r2_icache.sh > $O/voice_kernel_icache_counters.txt 2>&1; cat $O/voice_kernel_icache_counters.txt
