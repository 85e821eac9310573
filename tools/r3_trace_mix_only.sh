export TMPDIR=/tmp
O=gpurun_out/r3t2; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o p -- python tools/trace_mix_only.py > $O/prof.log 2>&1
python - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/r3t2/prof/**/p_kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("VoiceWave", "ApplyParams", "BusReduce", "PostSplit", "PostFir", "PostShift"):
        if k in n: return k
    return n[:24]
vi = [i for i, r in enumerate(rows) if "VoiceWave" in r["Kernel_Name"]]
i0 = vi[310]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + 32]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{short(r['Kernel_Name']):12s} q{r.get('Queue_Id','?'):>3s} start {s/1e3:8.1f} us  end {e/1e3:8.1f} us  dur {(e-s)/1e3:6.1f}")
PY
