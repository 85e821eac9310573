"""Reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv) and prints, for a few consecutive updates in the middle of the
run, every dispatch's start and end relative to the update's voice kernel (us): how the main stream (parameters, voices)
and the post stream (reduction, effects, post-process) of the pipelined update overlap.  usage: step_timeline.py TRACE.csv [n [fraction]]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
voice = [i for i, e in enumerate(ev) if ("VoiceWaveKernel" in e[2] or "VoiceWave16Kernel" in e[2])]
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5     # where in the run (by voice-kernel count) the window lies
mid = voice[int(len(voice) * frac)]
t0 = ev[mid][0]
last = voice[voice.index(mid) + n]
def short(k):
    k = k.split("(")[0]
    for p in ("void oalgpu::(anonymous namespace)::", "oalgpu::(anonymous namespace)::", "void oalgpu::", "void "):
        k = k.replace(p, "")
    return k[:48]
for s, e, k, q in ev[mid - 4:last + 1]:
    print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f}  {(e - s) / 1e3:7.2f} us  q{q:>3}  {short(k)}")
per = [ev[b][0] - ev[a][0] for a, b in zip(voice[len(voice) // 4:-2], voice[len(voice) // 4 + 1:-1])]
per.sort()
print(f"voice kernel period: median {per[len(per) // 2] / 1e3:.2f} us over {len(per)} updates (under the profiler)")
