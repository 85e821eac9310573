# Round-3 A/B of the HRTF voice kernel's FIR: the matrix pipe in split half precision (default) against packed
# fp32 VALU FMAs (bench.py --fir valu = OALGPU_CTX_FIR_VALU): parity of both forms, bench lines at the default and
# at the driver's K = 20, kernel stats, SQ / MFMA counters.  gpurun -- "bash tools/r3_ab.sh"
export TMPDIR=/tmp
O=gpurun_out/r3ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_baseline_configs.py tests/test_gpu_pipeline.py -m gpu -x -q > $O/pytest_variants.log 2>&1; tail -4 $O/pytest_variants.log
for fir in mfma valu mfma valu; do
  timeout 300 python bench.py --fir $fir --no-cpu-baseline < /dev/null > $O/bench_$fir.json 2> $O/bench_$fir.err
  timeout 300 python bench.py --fir $fir --no-cpu-baseline --steps 20 --warmup 5 < /dev/null > $O/bench20_$fir.json 2> $O/bench20_$fir.err
  python - <<PY
import json
for f in ("bench", "bench20"):
    try:
        d = json.load(open("$O/%s_$fir.json" % f))
        print("$fir", f, "%.1f us/step" % (d["ms_per_step"] * 1e3), "%.1f M voices/s" % (d["value"] / 1e6), "kernel %.1f us" % (d["roofline"]["kernel_ms"] * 1e3),
              "repeats", d["config"]["repeat_ms_per_step"], d["roofline"]["kernel"])
    except Exception as e:
        print("$fir", f, "ERR", e); print(open("$O/%s_$fir.err" % f).read()[-1500:])
PY
done
for fir in mfma valu; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$fir -o p -- python bench.py --fir $fir --steps 100 --warmup 5 --repeats 0 --no-cpu-baseline < /dev/null > $O/prof_$fir.log 2>&1
  cp $O/prof_$fir/p_kernel_stats.csv $O/kernel_stats_$fir.csv; head -6 $O/kernel_stats_$fir.csv | cut -c1-160
done
for fir in mfma valu; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY"; do
    tag=$(echo $set | cut -d' ' -f1)
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${fir}_$tag -o pmc -- python bench.py --fir $fir --steps 6 --warmup 2 --repeats 0 --no-cpu-baseline < /dev/null > $O/pmc_${fir}_$tag.log 2>&1
    python - <<PY >> $O/sq_counters.txt
import csv,collections
try:
    rows=[r for r in csv.DictReader(open("$O/pmc_${fir}_$tag/pmc_counter_collection.csv")) if "VoiceWave" in r["Kernel_Name"]]
    d=collections.defaultdict(list)
    for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in d.items(): print("$fir", k, len(v), sorted(v)[len(v)//2])
except Exception as e: print("ERR", "$fir", "$tag", e)
PY
  done
done
cat $O/sq_counters.txt
timeout 300 python tools/phase_times.py > $O/phase_times_mfma.txt 2>&1; tail -12 $O/phase_times_mfma.txt
timeout 300 python tools/phase_times.py 1 > $O/phase_times_valu.txt 2>&1; tail -12 $O/phase_times_valu.txt
