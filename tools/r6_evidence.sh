# Round-6 evidence for profiles/r6/evidence (ONE gpurun call): the GPU suite's log, bench lines (the driver's K = 20, default, every config,
# config 4 on the stream rows), rocprofv3 kernel stats of the same commands, FETCH_SIZE / WRITE_SIZE and SQ counter passes per config (every
# counter file carries the hash of the kernel sources it was collected on: bench.py refuses another tree's), the binding's end-to-end times.
#   gpurun --timeout 3000 -- "bash tools/r6_evidence.sh"      -> gpurun_out/r6e/
export TMPDIR=/tmp
O=gpurun_out/r6e
rm -rf $O; mkdir -p $O
lscpu | head -20 > $O/gpu_box_lscpu.txt
SRC_HASH=$(python -c "import bench; print(bench.kernel_sources_hash())")
echo "kernel sources sha256: $SRC_HASH" > $O/kernel_sources_hash.txt
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 < /dev/null > $O/bench_config3_driver_args.json 2> $O/bench.err; cut -c1-260 $O/bench_config3_driver_args.json
timeout 900 python bench.py < /dev/null > $O/bench_config3_default.json 2>> $O/bench.err; cut -c1-260 $O/bench_config3_default.json
# the driver's command under rocprofv3: the summary whose voice-kernel average must agree with the line's kernel_ms
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_k20 -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline < /dev/null > $O/prof_c3_k20.log 2>&1
cp $(find $O/prof_c3_k20 -name "p_kernel_stats.csv" | head -1) $O/config3_driver_args_kernel_stats.csv; head -5 $O/config3_driver_args_kernel_stats.csv | cut -c1-160
python tools/step_timeline.py $(find $O/prof_c3_k20 -name "p_kernel_trace.csv" | head -1) 4 0.2 > $O/step_timeline_config3_driver_args.txt 2>&1
run_config() {   # $1 config, $2 xflags, $3 tag, $4 more bench.py arguments
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$3 -o p -- python bench.py --config $1 --xflags $2 $4 --steps 100 --warmup 5 --repeats 0 --no-cpu-baseline < /dev/null > $O/prof_$3.log 2>&1
  cp $(find $O/prof_$3 -name "p_kernel_stats.csv" | head -1) $O/$3_kernel_stats.csv
  python tools/step_timeline.py $(find $O/prof_$3 -name "p_kernel_trace.csv" | head -1) 4 0.2 > $O/step_timeline_$3.txt 2>&1
  head -5 $O/$3_kernel_stats.csv | cut -c1-160
  timeout 300 python bench.py --config $1 --xflags $2 $4 --steps 300 --warmup 20 --no-cpu-baseline < /dev/null > $O/bench_$3.json 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$3_$ctr -o pmc -- python bench.py --config $1 --xflags $2 $4 --steps 20 --warmup 3 --repeats 0 --preroll 50 --no-cpu-baseline < /dev/null > $O/pmc_$3_$ctr.log 2>&1
  done
}
run_config 3 0 config3_launched "--resident off"     # (what the driver's K = 20 runs: a launch per update, VoiceWave16Kernel)
run_config 3 0 config3                               # (blocks >= 48 steps: the resident launch of voice_wave.hip's kernel)
run_config 2 0 config2
run_config 4 0 config4
run_config 4 8 config4_stream_rows
run_config 5 0 config5
SRC_HASH=$SRC_HASH python - <<'PY'
import csv, json, collections, glob, os
O = "gpurun_out/r6e"
H = os.environ["SRC_HASH"]
VK = ("VoiceWaveKernel", "VoiceWave16Kernel", "VoiceSliceKernel", "VoiceRowsKernel")
ALG = {3: 3956 + 384 + 16 + 512 + 512, 2: 3956 + 384 + 16 + 2 * 4 * 5 + 4 * 5, 4: 3956 + 384 + 16 + 3 * 4 * 5 + 2 * 3 * 4 * 4, 5: 3956 + 384 + 16 + 512 + 512 + 3 * 4 * 4}
out, traffic = {}, {"note": "HBM bytes per launch of the voice kernel of each BASELINE config, from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in "
    "separate passes, KiB medians over >= 20 launches; tools/r6_evidence.sh). hbm_bytes_per_launch = 2 x FETCH_SIZE + WRITE_SIZE: the gfx950 FETCH_SIZE "
    "correction of MI355X_MICROARCH.md (calibrated for 16 B/lane streams; these kernels read 4 B/lane rows, so this is an upper bound). "
    "algorithmic_bytes_per_launch: SURVEY.md 8(d) x voices.", "kernel_sources_sha256": H, "configs": {}}
for tag, c in (("config3_launched", 3), ("config3", 3), ("config2", 2), ("config4", 4), ("config4_stream_rows", 4), ("config5", 5)):
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
        if any(n in k for n in VK) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
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
for tag in ("config3_launched", "config3", "config2", "config4", "config4_stream_rows", "config5"):
    try:
        j = json.loads(open(f"{O}/bench_{tag}.json").read().strip().splitlines()[-1])
        print(tag, "ms_per_step %.5f value %.1fM kernel %s kernel_ms %.5f" % (j["ms_per_step"], j["value"] / 1e6, j["roofline"]["kernel"], j["roofline"]["kernel_ms"]))
    except Exception as e: print(tag, "bench failed", e)
PY
# SQ counters of the voice kernels of configs 3, 2, 4, 5
: > $O/voice_kernel_sq_counters.txt
for c in 3 2 4 5; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_FLAT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32"; do
    rm -rf $O/pmc_sq
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_sq -o pmc -- python bench.py --config $c --resident off --steps 6 --warmup 2 --repeats 0 --preroll 20 --no-cpu-baseline < /dev/null > $O/pmc_sq.log 2>&1
    python - >> $O/voice_kernel_sq_counters.txt <<PY
import csv, collections, glob
try:
    f = glob.glob("$O/pmc_sq/**/pmc_counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if any(n in r["Kernel_Name"] for n in ("VoiceWave", "VoiceRowsKernel", "VoiceSliceKernel"))]
    d = collections.defaultdict(list)
    for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    name = rows[0]["Kernel_Name"] if rows else "?"
    for k, v in d.items(): print($c, k, len(v), sorted(v)[len(v) // 2], name[name.find("Voice"):][:70])
except Exception as e: print("ERR", $c, e)
PY
  done
done
rm -rf $O/pmc_sq
SRC_HASH=$SRC_HASH python - <<'PY'
import json, collections, os
O = "gpurun_out/r6e"
cfg = collections.defaultdict(dict)
for line in open(f"{O}/voice_kernel_sq_counters.txt"):
    p = line.split(None, 4)
    if len(p) < 5 or p[0] == "ERR": continue
    c, name, calls, val, kern = p
    cfg[c][name] = float(val)
    k = kern.strip(); k = (k[:k.find(">") + 1] if ">" in k else k.split("(")[0]).replace("oalgpu::(anonymous namespace)::", "").replace("oalgpu::", "")
    cfg[c]["kernel_profiled"] = k
for c in cfg:
    cfg[c]["voices"] = 8192 if c == "4" else 4096
    try: cfg[c]["kernel"] = json.loads(open(f"{O}/bench_config{c}{'_launched' if c == '3' else ''}.json").read().strip().splitlines()[-1])["roofline"]["kernel"]
    except Exception: cfg[c]["kernel"] = cfg[c].get("kernel_profiled")
json.dump({"note": "rocprofv3 --pmc passes of `bench.py --config N` (tools/r6_evidence.sh): medians per launch of the config's voice kernel",
           "kernel_sources_sha256": os.environ["SRC_HASH"], "configs": cfg}, open(f"{O}/voice_kernel_sq_counters.json", "w"), indent=1)
for c in sorted(cfg): print(c, cfg[c].get("kernel"), "LDS insts", cfg[c].get("SQ_INSTS_LDS"), "VALU", cfg[c].get("SQ_INSTS_VALU"), "MFMA", cfg[c].get("SQ_INSTS_MFMA"))
PY
timeout 300 python tools/phase_times16.py > $O/voice_wave16_phase_times.txt 2>&1; tail -4 $O/voice_wave16_phase_times.txt
timeout 300 python tools/phase_times_rows.py > $O/voice_rows_phase_times.txt 2>&1; tail -3 $O/voice_rows_phase_times.txt
timeout 300 python tools/bridge_period.py --updates 60 > $O/bridge_e2e.txt 2>&1; cut -c1-150 $O/bridge_e2e.txt | grep -v "render times"
# delete the raw traces (tens of MB): the summaries above are what profiles/r6 keeps
rm -rf $O/prof_* $O/pmc_*/ 2>/dev/null; find $O -name "*.csv" -size +2M -delete; du -sh $O
