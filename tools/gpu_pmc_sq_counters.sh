# SQ counter passes of the voice kernel (profiles/r1p/voice_kernel_sq_counters.txt): gpurun -- "bash tools/gpu_pmc_sq_counters.sh"
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_sq
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_FLAT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_sq/pmc_$tag -o pmc -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline < /dev/null > gpurun_out/pmc_sq/pmc_$tag.log 2>&1
  python - <<PY
import csv,collections
try:
    rows=[r for r in csv.DictReader(open("gpurun_out/pmc_sq/pmc_$tag/pmc_counter_collection.csv")) if "VoiceWave" in r["Kernel_Name"]]
    d=collections.defaultdict(list)
    for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in d.items(): print(k, len(v), sorted(v)[len(v)//2])
except Exception as e: print("ERR", "$tag", e)
PY
done
