#!/bin/bash
# round-5 validation: the GPU suite, the bench line (K = 20 and defaults), the N = 2 path (two ranks on one device over the host
# transport: the re-exec under torch.distributed.run, the calibration), and the measurement build's tools
set -u
out=gpurun_out/${1:-r5val}
mkdir -p "$out"
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > "$out/pytest_gpu.log" 2>&1; tail -4 "$out/pytest_gpu.log"
timeout 600 python bench.py --steps 20 --warmup 5 > "$out/bench_k20.json" 2> "$out/bench_k20.err"; tail -3 "$out/bench_k20.err"
timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"; tail -3 "$out/bench_default.err"
timeout 600 python bench.py --gpus 2 --transport host --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_n2_host.json" 2> "$out/bench_n2_host.err"; tail -5 "$out/bench_n2_host.err"
timeout 300 python tools/phase_times.py > "$out/voice_kernel_phase_times.txt" 2>&1; tail -4 "$out/voice_kernel_phase_times.txt"
timeout 300 python tools/reverb_phase_times.py > "$out/reverb_phase_times.txt" 2>&1; tail -3 "$out/reverb_phase_times.txt"
python - "$out" <<'PY'
import json, sys
for n in ("bench_k20", "bench_default", "bench_n2_host"):
    try:
        j = json.loads(open(f"{sys.argv[1]}/{n}.json").read().strip().splitlines()[-1])
        r = j["roofline"]
        print(n, "n_gpus", j["n_gpus"], "ms_per_step %.5f value %.1fM kernel_ms %.5f frac %.3f mode: %s" % (j["ms_per_step"], j["value"] / 1e6, r["kernel_ms"], r["frac"], j["config"].get("voice_kernel_mode")))
        print("   comm", j["config"].get("comm"), "rccl_ranks", j["config"].get("rccl_ranks"), "calibration", j["config"].get("calibration"))
        print("   sources", r.get("traffic_source"), r.get("counters_source"), "event_floor", r.get("event_floor_ms"))
    except Exception as e:
        print(n, "failed:", e)
PY
