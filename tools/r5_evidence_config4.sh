# refresh of profiles/r5/evidence's config-4 files (after RebalanceWaveGroups and the slice kernel's prologue): bench line, rocprofv3 kernel
# stats, step timeline and FETCH_SIZE / WRITE_SIZE passes of both voice kernels.   gpurun -- "bash tools/r5_evidence_config4.sh" -> gpurun_out/r5e4/
export TMPDIR=/tmp
O=gpurun_out/r5e4
rm -rf $O; mkdir -p $O
run_config() {   # $1 config, $2 xflags, $3 tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$3 -o p -- python bench.py --config $1 --xflags $2 --steps 100 --warmup 5 --repeats 0 --no-cpu-baseline < /dev/null > $O/prof_$3.log 2>&1
  cp $(find $O/prof_$3 -name "p_kernel_stats.csv" | head -1) $O/$3_kernel_stats.csv
  python tools/step_timeline.py $(find $O/prof_$3 -name "p_kernel_trace.csv" | head -1) 4 0.2 > $O/step_timeline_$3.txt 2>&1
  head -4 $O/$3_kernel_stats.csv | cut -c1-160
  timeout 300 python bench.py --config $1 --xflags $2 --steps 300 --warmup 20 --no-cpu-baseline < /dev/null > $O/bench_$3.json 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$3_$ctr -o pmc -- python bench.py --config $1 --xflags $2 --steps 20 --warmup 3 --repeats 0 --preroll 50 --no-cpu-baseline < /dev/null > $O/pmc_$3_$ctr.log 2>&1
  done
}
run_config 4 0 config4
run_config 4 128 config4_slice_lines
python - <<'PY'
import csv, json, collections, glob
O = "gpurun_out/r5e4"
ALG4 = (3956 + 384 + 16 + 3 * 4 * 5 + 2 * 3 * 4 * 4) * 8192
res = {}
for tag in ("config4", "config4_slice_lines"):
    d = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"{O}/pmc_{tag}_{ctr}/**/pmc_counter_collection.csv", recursive=True)
        v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == ctr and ("VoiceWaveKernel" in r["Kernel_Name"] or "VoiceSliceKernel" in r["Kernel_Name"]))
        d[ctr] = v[len(v) // 2] * 1024
    j = json.loads(open(f"{O}/bench_{tag}.json").read().strip().splitlines()[-1])
    res[tag] = {"config": 4, "voices": 8192, "kernel": j["roofline"]["kernel"], "fetch_size_bytes_raw": d["FETCH_SIZE"], "write_size_bytes": d["WRITE_SIZE"],
                "hbm_bytes_per_launch": int(2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]), "algorithmic_bytes_per_launch": ALG4,
                "ratio_to_algorithmic": (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) / ALG4, "ms_per_step": j["ms_per_step"], "kernel_ms": j["roofline"]["kernel_ms"]}
    print(tag, res[tag])
json.dump(res, open(f"{O}/voice_kernel_traffic_config4.json", "w"), indent=1)
PY
rm -rf $O/prof_* $O/pmc_*/ $O/pmc_*.log; du -sh $O
