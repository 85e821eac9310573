# Round-5 evidence for profiles/r5 (ONE gpurun call): the GPU suite's log, bench lines (default, the driver's K = 20, every config, config 4
# on the slice kernel), rocprofv3 kernel stats of the same commands, FETCH_SIZE / WRITE_SIZE passes per config, the binding's end-to-end times.
#   gpurun --timeout 2400 -- "bash tools/r5_evidence.sh"      -> gpurun_out/r5e/
export TMPDIR=/tmp
O=gpurun_out/r5e
rm -rf $O; mkdir -p $O
lscpu | head -20 > $O/gpu_box_lscpu.txt
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 < /dev/null > $O/bench_config3_driver_args.json 2> $O/bench.err; cut -c1-260 $O/bench_config3_driver_args.json
timeout 900 python bench.py < /dev/null > $O/bench_config3_default.json 2>> $O/bench.err; cut -c1-260 $O/bench_config3_default.json
# the driver's command under rocprofv3: the summary whose voice-kernel average must agree with the line's kernel_ms
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_k20 -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $O/prof_c3_k20.log 2>&1
cp $(find $O/prof_c3_k20 -name "p_kernel_stats.csv" | head -1) $O/config3_driver_args_kernel_stats.csv; head -5 $O/config3_driver_args_kernel_stats.csv | cut -c1-160
run_config() {   # $1 config, $2 xflags, $3 tag
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$3 -o p -- python bench.py --config $1 --xflags $2 --steps 100 --warmup 5 --repeats 0 --no-cpu-baseline < /dev/null > $O/prof_$3.log 2>&1
  cp $(find $O/prof_$3 -name "p_kernel_stats.csv" | head -1) $O/$3_kernel_stats.csv
  python tools/step_timeline.py $(find $O/prof_$3 -name "p_kernel_trace.csv" | head -1) 4 0.2 > $O/step_timeline_$3.txt 2>&1
  head -5 $O/$3_kernel_stats.csv | cut -c1-160
  timeout 300 python bench.py --config $1 --xflags $2 --steps 300 --warmup 20 --no-cpu-baseline < /dev/null > $O/bench_$3.json 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$3_$ctr -o pmc -- python bench.py --config $1 --xflags $2 --steps 20 --warmup 3 --repeats 0 --preroll 50 --no-cpu-baseline < /dev/null > $O/pmc_$3_$ctr.log 2>&1
  done
}
run_config 3 0 config3
run_config 2 0 config2
run_config 4 0 config4
run_config 4 128 config4_slice_lines
run_config 5 0 config5
python - <<'PY'
import csv, json, collections, glob
O = "gpurun_out/r5e"
ALG = {3: 3956 + 384 + 16 + 512 + 512, 2: 3956 + 384 + 16 + 2 * 4 * 5 + 4 * 5, 4: 3956 + 384 + 16 + 3 * 4 * 5 + 2 * 3 * 4 * 4, 5: 3956 + 384 + 16 + 512 + 512 + 3 * 4 * 4}
out, traffic = {}, {"note": "HBM bytes per launch of the voice kernel of each BASELINE config, from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in "
    "separate passes, KiB medians over >= 20 launches). hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE: the gfx950 FETCH_SIZE correction of "
    "MI355X_MICROARCH.md (calibrated for 16 B/lane streams; these kernels read 4 B/lane rows, so this is an upper bound). "
    "algorithmic_bytes_per_launch: SURVEY.md 8(d) x voices.", "configs": {}}
for tag, c in (("config3", 3), ("config2", 2), ("config4", 4), ("config4_slice_lines", 4), ("config5", 5)):
    per = collections.defaultdict(dict)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"{O}/pmc_{tag}_{ctr}/**/pmc_counter_collection.csv", recursive=True)
        if not fs: continue
        byk = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] == ctr: byk[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in byk.items():
            v.sort(); per[k][ctr] = {"median_kb": v[len(v) // 2], "calls": len(v)}
    out[tag] = per
    for k, d in per.items():
        if ("VoiceWaveKernel" in k or "VoiceSliceKernel" in k) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            voices = 8192 if c == 4 else 4096
            f, w = d["FETCH_SIZE"]["median_kb"] * 1024, d["WRITE_SIZE"]["median_kb"] * 1024
            try: bname = json.loads(open(f"{O}/bench_{tag}.json").read().strip().splitlines()[-1])["roofline"]["kernel"]
            except Exception: bname = k[:80]
            traffic["configs"][tag] = {"config": c, "voices": voices, "kernel": bname, "fetch_size_bytes_raw": f, "write_size_bytes": w,
                                       "hbm_bytes_per_launch": int(2 * f + w), "algorithmic_bytes_per_launch": ALG[c] * voices,
                                       "ratio_to_algorithmic": (2 * f + w) / (ALG[c] * voices)}
            print(tag, bname, "FETCH", round(f / 1e6, 1), "WRITE", round(w / 1e6, 1), "2F+W", round((2 * f + w) / 1e6, 1), "alg", round(ALG[c] * voices / 1e6, 1),
                  "ratio %.2f" % ((2 * f + w) / (ALG[c] * voices)))
json.dump(out, open(f"{O}/pmc_hbm_by_kernel.json", "w"), indent=1)
json.dump(traffic, open(f"{O}/voice_kernel_traffic.json", "w"), indent=1)
for tag in ("config3", "config2", "config4", "config4_slice_lines", "config5"):
    try:
        j = json.loads(open(f"{O}/bench_{tag}.json").read().strip().splitlines()[-1])
        print(tag, "ms_per_step %.5f value %.1fM kernel %s kernel_ms %.5f" % (j["ms_per_step"], j["value"] / 1e6, j["roofline"]["kernel"], j["roofline"]["kernel_ms"]))
    except Exception as e: print(tag, "bench failed", e)
PY
timeout 300 python tools/bridge_period.py --updates 60 > $O/bridge_e2e.txt 2>&1; cut -c1-150 $O/bridge_e2e.txt | grep -v "render times"
# delete the raw traces (tens of MB): the summaries above are what profiles/r5 keeps
rm -rf $O/prof_* $O/pmc_*/ 2>/dev/null; find $O -name "*.csv" -size +2M -delete; du -sh $O
