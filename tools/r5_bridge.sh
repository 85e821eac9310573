#!/bin/bash
set -u
out=gpurun_out/${1:-r5br}
mkdir -p "$out"
( time timeout 900 python -m pytest tests/test_bridge.py tests/test_buffer_lifetime.py tests/test_gpu_resident.py -q -m gpu ) > "$out/tests.log" 2>&1; tail -6 "$out/tests.log"
timeout 600 python tools/bridge_period.py > "$out/bridge_e2e.txt" 2>&1; cat "$out/bridge_e2e.txt"
