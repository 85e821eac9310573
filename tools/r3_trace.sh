# kernel timeline of a few steady-state updates (rocprofv3 kernel trace).  gpurun -- "bash tools/r3_trace.sh [bench args]"
export TMPDIR=/tmp
O=gpurun_out/r3t; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof -o p -- python bench.py --steps 60 --warmup 10 --repeats 0 --no-cpu-baseline "$@" < /dev/null > $O/prof.log 2>&1
python - <<'PY'
import csv
import glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/r3t/prof/**/p_kernel_trace.csv", recursive=True)[0])))
for c in glob.glob("gpurun_out/r3t/prof/**/p_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(c)):
        rows.append({"Kernel_Name": "COPY " + r.get("Direction", ""), "Start_Timestamp": r["Start_Timestamp"], "End_Timestamp": r["End_Timestamp"], "Queue_Id": "-"})
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("VoiceWave", "StageParams", "ApplyParams", "BusReduce", "PostDirect"):
        if k in n: return k
    return n[:24]
# find the 300th voice kernel and print 40 kernels from there
vi = [i for i, r in enumerate(rows) if "VoiceWave" in r["Kernel_Name"]]
i0 = vi[min(300, len(vi) - 12)]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 36]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{short(r['Kernel_Name']):12s} q{r.get('Queue_Id','?'):>3s} start {s/1e3:8.1f} us  end {e/1e3:8.1f} us  dur {(e-s)/1e3:6.1f}")
PY
