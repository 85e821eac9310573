# SQ counter passes of config 4's voice kernel (slice kernel by default; XF=8: stream rows): gpurun -- "bash tools/r5_slice_pmc.sh"
export TMPDIR=/tmp
XF=${XF:-0}
OUT=gpurun_out/pmc_slice_x$XF
mkdir -p $OUT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_FLAT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$tag -o pmc -- python bench.py --config 4 --steps 6 --warmup 2 --preroll 4 --no-cpu-baseline --repeats 0 --xflags $XF < /dev/null > $OUT/pmc_$tag.log 2>&1
  python - <<PY
import csv,collections,glob
try:
    f=glob.glob("$OUT/pmc_$tag/**/*counter_collection.csv", recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if "Voice" in r["Kernel_Name"] and "Kernel<" in r["Kernel_Name"]]
    d=collections.defaultdict(list)
    for r in rows: d[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in d.items(): print(k[1], len(v), sorted(v)[len(v)//2], k[0])
except Exception as e: print("ERR", "$tag", e)
PY
done
