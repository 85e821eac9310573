# The GPU-side command behind profiles/r1p: python bench.py (the bench line), rocprofv3 kernel stats and the two
# HBM PMC passes of config 3.  Run from the repo root on the GPU box: gpurun -- "bash tools/gpu_profile.sh"
export TMPDIR=/tmp
mkdir -p gpurun_out/r1p
timeout 600 python bench.py < /dev/null > gpurun_out/r1p/bench.json 2> gpurun_out/r1p/bench.err
cat gpurun_out/r1p/bench.json | cut -c1-400
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1p/prof_c3 -o p -- python bench.py --steps 60 --warmup 5 --no-cpu-baseline < /dev/null > gpurun_out/r1p/prof_c3.log 2>&1
head -4 gpurun_out/r1p/prof_c3/p_kernel_stats.csv | cut -c1-160
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r1p/pmc_$c -o pmc -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline < /dev/null > gpurun_out/r1p/pmc_$c.log 2>&1
done
python - <<'PY'
import csv
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows=[r for r in csv.DictReader(open(f"gpurun_out/r1p/pmc_{c}/pmc_counter_collection.csv")) if "VoiceWave" in r["Kernel_Name"] and r["Counter_Name"]==c]
    vals=sorted(float(r["Counter_Value"]) for r in rows)
    print(c, len(vals), "median", vals[len(vals)//2])
PY
for c in 2 4 5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1p/prof_c$c -o p -- python bench.py --config $c --steps 60 --warmup 5 --no-cpu-baseline < /dev/null > gpurun_out/r1p/prof_c$c.log 2>&1
  timeout 300 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline < /dev/null > gpurun_out/r1p/bench_c$c.json 2>/dev/null
done
timeout 300 python tools/step_period.py < /dev/null 2>&1 | grep period > gpurun_out/r1p/step_period.txt
timeout 200 python tools/phase_times.py < /dev/null 2>&1 | tail -24 > gpurun_out/r1p/phase_times.txt
