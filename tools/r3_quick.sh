# quick iteration: parity of the voice-kernel forms, bench A/B, phase stamps.  gpurun -- "bash tools/r3_quick.sh"
export TMPDIR=/tmp
O=gpurun_out/r3q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_baseline_configs.py tests/test_gpu_pipeline.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for fir in mfma valu mfma valu; do
  timeout 300 python bench.py --fir $fir --no-cpu-baseline < /dev/null > $O/bench_$fir.json 2> $O/bench_$fir.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$fir.json"))
    print("$fir", "%.1f us/step" % (d["ms_per_step"] * 1e3), "%.1f M voices/s" % (d["value"] / 1e6), "kernel %.1f us" % (d["roofline"]["kernel_ms"] * 1e3), "repeats med %.1f" % (d["config"]["repeat_ms_per_step"]["median"] * 1e3))
except Exception as e:
    print("$fir ERR", e); print(open("$O/bench_$fir.err").read()[-1500:])
PY
done
timeout 300 python tools/phase_times.py > $O/phase_times_mfma.txt 2>&1; tail -9 $O/phase_times_mfma.txt
