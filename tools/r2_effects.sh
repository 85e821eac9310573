#!/bin/bash
# per-kernel time of the EffectState kernels: bash tools/r2_effects.sh (on the GPU box, writes gpurun_out/r2/effects_kernel_stats.csv)
set -e
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p "$ROOT/gpurun_out/r2"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fxprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fxprof -o fx -- python "$ROOT/tools/effects_time.py" > /tmp/fxprof.log 2>&1 || { tail -20 /tmp/fxprof.log; exit 1; }
F=$(find /tmp/fxprof -name "*kernel_stats.csv" | head -1)
cp "$F" "$ROOT/gpurun_out/r2/effects_kernel_stats.csv"
cut -c1-200 "$F"
