# full GPU suite + the bench line at the default and at the driver's K = 20.  gpurun -- "bash tools/r3_full.sh"
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; tail -6 $O/pytest_all.log
for args in "" "--steps 20 --warmup 5" "--fir valu"; do
  tag=$(echo "default$args" | tr -d ' -')
  timeout 300 python bench.py $args --no-cpu-baseline < /dev/null > $O/bench_$tag.json 2> $O/bench_$tag.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$tag.json"))
    print("$tag", "%.1f us/step" % (d["ms_per_step"] * 1e3), "%.1f M voices/s" % (d["value"] / 1e6), "kernel %.1f us" % (d["roofline"]["kernel_ms"] * 1e3), "repeats med %.1f" % (d["config"]["repeat_ms_per_step"]["median"] * 1e3), "e2e %.3f ms" % d["config"]["e2e_ms_per_update"])
except Exception as e:
    print("$tag ERR", e); print(open("$O/bench_$tag.err").read()[-1500:])
PY
done
