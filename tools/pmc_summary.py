"""Summarise a rocprofv3 --pmc counter_collection.csv: mean per dispatch and per voice for kernels matching a substring."""
import csv, collections, sys
f, pat = sys.argv[1], sys.argv[2]
nv = float(sys.argv[3]) if len(sys.argv) > 3 else 4096.0
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if pat in k:
        print(k[:90])
        for c, v in sorted(d.items()):
            m = sum(v) / len(v)
            print(f"   {c:26s} mean={m:14.1f}  per voice {m / nv:10.1f}")
