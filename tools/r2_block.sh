# Round-2: the workgroup-per-voice kernel (voice_block.hip) against the wavefront kernel.
# gpurun -- "bash tools/r2_block.sh"; results under gpurun_out/r2blk.
export TMPDIR=/tmp
O=gpurun_out/r2blk
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_block4.log 2>&1; tail -5 $O/pytest_block4.log
OALGPU_BLOCK_WAVES=3 timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_block3.log 2>&1; tail -3 $O/pytest_block3.log
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline < /dev/null > $O/bench_$tag.json 2> $O/bench_$tag.err
  python -c "import json;d=json.load(open('$O/bench_$tag.json'));print('$tag', round(d['value']/1e6,2),'Mv/s step', round(d['ms_per_step']*1e3,2),'us kernel', round(d['roofline']['kernel_ms']*1e3,2), d['roofline']['kernel'])" || tail -3 $O/bench_$tag.err
}
run wave_valu OALGPU_VOICE_KERNEL=wave OALGPU_FIR=valu
run wave_mfma OALGPU_VOICE_KERNEL=wave
run block4 OALGPU_BLOCK_WAVES=4
run block3 OALGPU_BLOCK_WAVES=3
for tag in block4 block3; do
  w=${tag#block}
  OALGPU_BLOCK_WAVES=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o p -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline < /dev/null > $O/prof_$tag.log 2>&1
  head -5 $O/prof_$tag/p_kernel_stats.csv | cut -c1-180
done
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -o pmc -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline < /dev/null > $O/pmc_$tag.log 2>&1
  python - <<PY
import csv,collections
try:
    rows=[r for r in csv.DictReader(open("$O/pmc_$tag/pmc_counter_collection.csv")) if "VoiceBlock" in r["Kernel_Name"]]
    d=collections.defaultdict(list)
    for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in d.items(): print("block4", k, len(v), sorted(v)[len(v)//2])
except Exception as e: print("ERR", "$tag", e)
PY
done 2>&1 | tee $O/pmc_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline < /dev/null > $O/pmc_$c.log 2>&1
  python - <<PY
import csv
rows=[r for r in csv.DictReader(open("$O/pmc_$c/pmc_counter_collection.csv")) if "VoiceBlock" in r["Kernel_Name"] and r["Counter_Name"]=="$c"]
vals=sorted(float(r["Counter_Value"]) for r in rows)
print("$c", len(vals), "median", vals[len(vals)//2] if vals else None)
PY
done 2>&1 | tee -a $O/pmc_summary.txt
