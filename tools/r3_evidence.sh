# Round-3 evidence for profiles/r3 (ONE gpurun call): the default bench line and the driver's (K = 20), rocprofv3 kernel stats
# + bench line + FETCH_SIZE / WRITE_SIZE passes per BASELINE config, SQ / MFMA / instruction-cache counters and phase stamps
# of the HRTF voice kernel, the packed-fp32 op_sel microbenchmark and the biquad scan beside an MFMA partner.
#   gpurun --timeout 2400 -- "bash tools/r3_evidence.sh"      -> gpurun_out/r3e/
export TMPDIR=/tmp
O=gpurun_out/r3e
mkdir -p $O
lscpu | head -20 > $O/gpu_box_lscpu.txt
timeout 900 python bench.py < /dev/null > $O/bench_config3_default.json 2> $O/bench.err; cut -c1-400 $O/bench_config3_default.json
timeout 300 python bench.py --steps 20 --warmup 5 < /dev/null > $O/bench_config3_driver_args.json 2>/dev/null
timeout 300 python bench.py --fir valu --no-cpu-baseline < /dev/null > $O/bench_config3_fir_valu.json 2>/dev/null
for c in 3 2 4 5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c$c -o p -- python bench.py --config $c --steps 100 --warmup 5 --repeats 0 --no-cpu-baseline < /dev/null > $O/prof_c$c.log 2>&1
  cp $(find $O/prof_c$c -name "p_kernel_stats.csv" | head -1) $O/config${c}_kernel_stats.csv
  head -7 $O/config${c}_kernel_stats.csv | cut -c1-160
  timeout 300 python bench.py --config $c --steps 300 --warmup 20 --no-cpu-baseline < /dev/null > $O/bench_config$c.json 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_c${c}_$ctr -o pmc -- python bench.py --config $c --steps 20 --warmup 3 --repeats 0 --preroll 50 --no-cpu-baseline < /dev/null > $O/pmc_c${c}_$ctr.log 2>&1
  done
done
python - <<'PY'
import csv, json, collections, glob
out = {}
for c in (2, 3, 4, 5):
    per = collections.defaultdict(dict)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"gpurun_out/r3e/pmc_c{c}_{ctr}/**/pmc_counter_collection.csv", recursive=True)
        if not fs: continue
        rows = [r for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == ctr]
        byk = collections.defaultdict(list)
        for r in rows: byk[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in byk.items():
            v.sort(); per[k][ctr] = {"median_kb": v[len(v) // 2], "calls": len(v)}
    out[c] = per
json.dump(out, open("gpurun_out/r3e/pmc_hbm_by_kernel.json", "w"), indent=1)
for c, per in out.items():
    for k, d in per.items():
        if any(x in k for x in ("VoiceWave", "Conv", "Reverb", "BusReduce", "Post")):
            print(c, k[:80], {n: round(x["median_kb"]) for n, x in d.items()})
PY
# SQ counters of the HRTF voice kernel, both FIR forms
for fir in mfma valu; do
  : > $O/voice_kernel_sq_counters_$fir.txt
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_FLAT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf $O/pmc_sq
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_sq -o pmc -- python bench.py --fir $fir --steps 6 --warmup 2 --repeats 0 --preroll 20 --no-cpu-baseline < /dev/null > $O/pmc_sq.log 2>&1
    python - >> $O/voice_kernel_sq_counters_$fir.txt <<PY
import csv, collections, glob
try:
    f = glob.glob("$O/pmc_sq/**/pmc_counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "VoiceWave" in r["Kernel_Name"]]
    d = collections.defaultdict(list)
    for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in d.items(): print(k, len(v), sorted(v)[len(v) // 2])
except Exception as e: print("ERR", "$tag", e)
PY
  done
  echo "== $fir"; cat $O/voice_kernel_sq_counters_$fir.txt
done
rm -rf $O/pmc_sq
timeout 300 python tools/phase_times.py > $O/voice_kernel_phase_times.txt 2>&1; tail -4 $O/voice_kernel_phase_times.txt
timeout 60 ./tools/ubench_pk_opsel > $O/ubench_pk_opsel.txt 2>&1; tail -14 $O/ubench_pk_opsel.txt
for m in 0 1 2; do timeout 60 ./tools/ubench_bqscan $m | grep -v "^  wave"; done > $O/ubench_bqscan.txt 2>&1
timeout 120 python tools/host_submit_time.py > $O/host_submit_time.txt 2>&1; tail -6 $O/host_submit_time.txt
bash tools/r3_trace.sh > $O/step_timeline.txt 2>&1; head -20 $O/step_timeline.txt
du -sh $O; find $O -name "*.csv" -size +2M -delete; rm -rf $O/prof_c* $O/pmc_c*
