# instruction-cache behaviour of the voice kernel (profiles/r2/voice_kernel_icache_counters.txt)
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_ic
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_ic/pmc_$tag -o pmc -- python bench.py --steps 6 --warmup 2 --repeats 0 --no-cpu-baseline < /dev/null > gpurun_out/pmc_ic/pmc_$tag.log 2>&1
  python - <<PY
import csv,collections
try:
    rows=[r for r in csv.DictReader(open("gpurun_out/pmc_ic/pmc_$tag/pmc_counter_collection.csv")) if "VoiceWave" in r["Kernel_Name"]]
    d=collections.defaultdict(list)
    for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in d.items(): print(k, len(v), sorted(v)[len(v)//2])
except Exception as e: print("ERR", "$tag", e)
PY
done
