"""Reads a rocprofv3 kernel trace (sqlite, --kernel-trace) and prints, for the LAST resident launch of the voice kernel, when the
update's reduction and post-process started and ended relative to the launch's start.
usage: python tools/trace_kernels.py <trace_results.db>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, display_name from {ks}")}


def short(n):
    m = re.search(r"(\w+Kernel)", n)
    return m.group(1) if m else n[:40]


rows = [(short(names[k]), s, e) for k, s, e in cur.execute(f"select kernel_id, start, end from {kd} order by start")]
voice = [r for r in rows if r[0] == "VoiceWaveKernel" and r[2] - r[1] > 300_000]       # resident launches last for hundreds of us
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
v = voice[which]
print(f"resident launches in the trace: {len(voice)}; this one lasted {(v[2] - v[1]) / 1e3:.1f} us")
red = [r for r in rows if r[0] == "BusReduceResidentKernel" and v[1] - 50_000 < r[1] < v[2] + 200_000]
post = [r for r in rows if r[0] == "PostResidentKernel" and v[1] - 50_000 < r[1] < v[2] + 200_000]
for i, (r, p) in enumerate(zip(red, post)):
    print(f"update {i:3d}: reduction [{(r[1] - v[1]) / 1e3:8.1f} .. {(r[2] - v[1]) / 1e3:8.1f}] {(r[2] - r[1]) / 1e3:6.1f} us   "
          f"post-process [{(p[1] - v[1]) / 1e3:8.1f} .. {(p[2] - v[1]) / 1e3:8.1f}] {(p[2] - p[1]) / 1e3:6.1f} us")
