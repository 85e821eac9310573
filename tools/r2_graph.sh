# oalgpu_update_graph_* A/B: the default step loop against hipGraphs of 20 updates, configs 3 and 2,
# and the host's share of a step in both (tools/host_submit_time.py)
export TMPDIR=/tmp
O=gpurun_out/graph; mkdir -p $O
for c in 3 2; do
  for g in 0 20; do
    timeout 300 python bench.py --config $c --graph $g --no-cpu-baseline < /dev/null > $O/bench_c${c}_g$g.json 2>$O/bench_c${c}_g$g.err
    python - <<PY
import json
d=json.load(open("$O/bench_c${c}_g$g.json"))
print("config $c graph $g: %.1f us/step" % (d["ms_per_step"]*1000), "%.1f M voices/s" % (d["value"]/1e6), d["config"].get("repeat_ms_per_step"))
PY
  done
done
timeout 300 python tools/host_submit_time.py 2>&1 | tail -6
