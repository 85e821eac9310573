# Round-2 A/B of the HRTF FIR: packed-VALU (OALGPU_FIR=valu) against the matrix-pipe Toeplitz form
# (default).  gpurun -- "bash tools/r2_ab.sh"; results under gpurun_out/r2ab.
export TMPDIR=/tmp
O=gpurun_out/r2ab
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for fir in valu mfma; do
  OALGPU_FIR=$fir timeout 300 python bench.py --no-cpu-baseline < /dev/null > $O/bench_$fir.json 2> $O/bench_$fir.err
  python -c "import json;d=json.load(open('$O/bench_$fir.json'));print('$fir', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel'])"
done
for fir in valu mfma; do
  OALGPU_FIR=$fir timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$fir -o p -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline < /dev/null > $O/prof_$fir.log 2>&1
  head -4 $O/prof_$fir/p_kernel_stats.csv | cut -c1-200
done
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  for fir in valu mfma; do
    OALGPU_FIR=$fir timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${fir}_$tag -o pmc -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline < /dev/null > $O/pmc_${fir}_$tag.log 2>&1
    python - <<PY
import csv,collections
try:
    rows=[r for r in csv.DictReader(open("$O/pmc_${fir}_$tag/pmc_counter_collection.csv")) if "VoiceWave" in r["Kernel_Name"]]
    d=collections.defaultdict(list)
    for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in d.items(): print("$fir", k, len(v), sorted(v)[len(v)//2])
except Exception as e: print("ERR", "$fir", "$tag", e)
PY
  done
done 2>&1 | tee $O/pmc_summary.txt
