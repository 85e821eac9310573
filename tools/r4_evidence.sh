# Round-4 evidence for profiles/r4 (ONE gpurun call): bench lines (default, the driver's K = 20, every config), rocprofv3 kernel
# stats and FETCH_SIZE / WRITE_SIZE passes per BASELINE config, SQ / MFMA / LDS counters of the voice kernels, phase stamps,
# the step's anatomy (tools/step_period.py, post_period.py, step_timeline.py) and the A/B runs DESIGN.md quotes.
#   gpurun --timeout 2400 -- "bash tools/r4_evidence.sh"      -> gpurun_out/r4e/
export TMPDIR=/tmp
O=gpurun_out/r4e
rm -rf $O; mkdir -p $O
lscpu | head -20 > $O/gpu_box_lscpu.txt
timeout 900 python bench.py < /dev/null > $O/bench_config3_default.json 2> $O/bench.err; cut -c1-300 $O/bench_config3_default.json
timeout 300 python bench.py --steps 20 --warmup 5 < /dev/null > $O/bench_config3_driver_args.json 2>/dev/null
# A/B on this box: the parameter block installed by the voice kernel's own wavefronts (OALGPU_CTX_APPLY_IN_VOICE_KERNEL = 16), stream
# rows instead of line accumulators (OALGPU_CTX_STREAM_ROWS = 8)
timeout 300 python bench.py --xflags 16 --no-cpu-baseline < /dev/null > $O/bench_config3_apply_in_voice_kernel.json 2>/dev/null
for c in 2 5; do timeout 300 python bench.py --config $c --xflags 8 --no-cpu-baseline < /dev/null > $O/bench_config${c}_stream_rows.json 2>/dev/null; done
for c in 3 2 4 5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c$c -o p -- python bench.py --config $c --steps 100 --warmup 5 --repeats 0 --no-cpu-baseline < /dev/null > $O/prof_c$c.log 2>&1
  cp $(find $O/prof_c$c -name "p_kernel_stats.csv" | head -1) $O/config${c}_kernel_stats.csv
  if [ $c = 3 ]; then python tools/step_timeline.py $(find $O/prof_c$c -name "p_kernel_trace.csv" | head -1) 4 0.2 > $O/step_timeline.txt 2>&1; fi
  if [ $c = 4 ] || [ $c = 5 ]; then python tools/step_timeline.py $(find $O/prof_c$c -name "p_kernel_trace.csv" | head -1) 4 0.2 > $O/step_timeline_config$c.txt 2>&1; fi
  head -6 $O/config${c}_kernel_stats.csv | cut -c1-150
  timeout 300 python bench.py --config $c --steps 300 --warmup 20 --no-cpu-baseline < /dev/null > $O/bench_config$c.json 2>/dev/null
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_c${c}_$ctr -o pmc -- python bench.py --config $c --steps 20 --warmup 3 --repeats 0 --preroll 50 --no-cpu-baseline < /dev/null > $O/pmc_c${c}_$ctr.log 2>&1
  done
done
python - <<'PY'
import csv, json, collections, glob
O = "gpurun_out/r4e"
out = {}
for c in (2, 3, 4, 5):
    per = collections.defaultdict(dict)
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"{O}/pmc_c{c}_{ctr}/**/pmc_counter_collection.csv", recursive=True)
        if not fs: continue
        rows = [r for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == ctr]
        byk = collections.defaultdict(list)
        for r in rows: byk[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in byk.items():
            v.sort(); per[k][ctr] = {"median_kb": v[len(v) // 2], "calls": len(v)}
    out[c] = per
json.dump(out, open(f"{O}/pmc_hbm_by_kernel.json", "w"), indent=1)
# the voice kernel's line per config, in the shape bench.py reads (profiles/voice_kernel_traffic.json)
ALG = {3: 3956 + 384 + 16 + 512 + 512, 2: 3956 + 384 + 16 + 2 * 4 * 5 + 4 * 5, 4: 3956 + 384 + 16 + 3 * 4 * 5 + 2 * 3 * 4 * 4, 5: 3956 + 384 + 16 + 512 + 512 + 3 * 4 * 4}
traffic = {"note": "HBM bytes per launch of the voice kernel of each BASELINE config, from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in "
           "separate passes, profiles/r4/pmc_hbm_by_kernel.json, KiB medians over >= 20 launches). hbm_bytes_per_launch = 2 x FETCH_SIZE + "
           "WRITE_SIZE: the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (calibrated for 16 B/lane streams; these kernels read 4 B/lane "
           "rows, so this is an upper bound). algorithmic_bytes_per_launch: SURVEY.md 8(d) x voices.", "configs": {}}
for c, per in out.items():
    for k, d in per.items():
        if "VoiceWaveKernel" in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            name = k[k.find("VoiceWaveKernel"):]
            name = name[:name.find(">") + 1].replace("oalgpu::(anonymous namespace)::", "").replace("oalgpu::", "")
            voices = 8192 if c == 4 else 4096
            f, w = d["FETCH_SIZE"]["median_kb"] * 1024, d["WRITE_SIZE"]["median_kb"] * 1024
            try: bname = json.loads(open(f"{O}/bench_config{c}.json").read().strip().splitlines()[-1])["roofline"]["kernel"]
            except Exception: bname = name
            traffic["configs"][str(c)] = {"config": c, "voices": voices, "kernel": bname, "kernel_profiled": name, "fetch_size_bytes_raw": f, "write_size_bytes": w,
                                          "hbm_bytes_per_launch": int(2 * f + w), "algorithmic_bytes_per_launch": ALG[c] * voices,
                                          "ratio_to_algorithmic": (2 * f + w) / (ALG[c] * voices)}
            print(c, name, "FETCH", round(f / 1e6, 1), "WRITE", round(w / 1e6, 1), "2F+W", round((2 * f + w) / 1e6, 1), "alg", round(ALG[c] * voices / 1e6, 1),
                  "ratio %.2f" % ((2 * f + w) / (ALG[c] * voices)))
json.dump(traffic, open(f"{O}/voice_kernel_traffic.json", "w"), indent=1)
PY
# SQ counters of the voice kernels of configs 3, 2, 5
: > $O/voice_kernel_sq_counters.txt
for c in 3 2 5; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_FLAT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
    rm -rf $O/pmc_sq
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_sq -o pmc -- python bench.py --config $c --steps 6 --warmup 2 --repeats 0 --preroll 20 --no-cpu-baseline < /dev/null > $O/pmc_sq.log 2>&1
    python - >> $O/voice_kernel_sq_counters.txt <<PY
import csv, collections, glob
try:
    f = glob.glob("$O/pmc_sq/**/pmc_counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "VoiceWave" in r["Kernel_Name"]]
    d = collections.defaultdict(list)
    for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    name = rows[0]["Kernel_Name"] if rows else "?"
    for k, v in d.items(): print($c, k, len(v), sorted(v)[len(v) // 2], name[name.find("VoiceWaveKernel"):][:70])
except Exception as e: print("ERR", $c, e)
PY
  done
done
rm -rf $O/pmc_sq
python - <<'PY'
import json, collections
O = "gpurun_out/r4e"
cfg = collections.defaultdict(dict)
for line in open(f"{O}/voice_kernel_sq_counters.txt"):
    p = line.split(None, 4)
    if len(p) < 5 or p[0] == "ERR": continue
    c, name, calls, val, kern = p
    cfg[c][name] = float(val)
    k = kern.strip(); k = k[:k.find(">") + 1].replace("oalgpu::(anonymous namespace)::", "").replace("oalgpu::", "")
    cfg[c]["kernel_profiled"] = k
for c in cfg:
    cfg[c]["voices"] = 8192 if c == "4" else 4096
    try: cfg[c]["kernel"] = json.loads(open(f"{O}/bench_config{c}.json").read().strip().splitlines()[-1])["roofline"]["kernel"]
    except Exception: cfg[c]["kernel"] = cfg[c].get("kernel_profiled")
json.dump({"note": "rocprofv3 --pmc passes of `bench.py --config N` (tools/r4_evidence.sh): medians per launch of the config's voice kernel", "configs": cfg},
          open(f"{O}/voice_kernel_sq_counters.json", "w"), indent=1)
PY
timeout 300 python tools/phase_times.py > $O/voice_kernel_phase_times.txt 2>&1; tail -4 $O/voice_kernel_phase_times.txt
for c in 2 4 5; do timeout 300 python tools/phase_times_lines.py $c > $O/phase_times_config$c.txt 2>&1; done
timeout 300 python tools/step_period.py 0 > $O/step_period.txt 2>&1; timeout 300 python tools/step_period.py 16 >> $O/step_period.txt 2>&1; cat $O/step_period.txt
timeout 120 python tools/post_period.py > $O/post_period.txt 2>&1; cat $O/post_period.txt
# configs 2, 4, 5: the two-stream pipeline against one stream (OALGPU_CTX_SERIAL = 4)
: > $O/step_period_configs_final.txt
for a in "0 2" "4 2" "0 4" "4 4" "0 5" "4 5"; do timeout 200 python tools/step_period.py $a >> $O/step_period_configs_final.txt 2>&1; done
cat $O/step_period_configs_final.txt
du -sh $O; find $O -name "*.csv" -size +2M -delete; rm -rf $O/prof_c* $O/pmc_c*
