#!/bin/bash
# machine LICM off (the build) against on (tools/ab/liboalgpu_licm_on.so) for the launched voice kernels: configs 3, 2, 5, 4, and
# config 3 with the parameter block installed by the voice kernel's epilogue (xflags 16), one gpurun call
set -u
out=gpurun_out/${1:-r5w}
mkdir -p "$out"
cp openal-soft_amd/liboalgpu.so /tmp/lib_off.so
for lib in off on off on; do
  if [ $lib = on ]; then cp tools/ab/liboalgpu_licm_on.so openal-soft_amd/liboalgpu.so; else cp /tmp/lib_off.so openal-soft_amd/liboalgpu.so; fi
  for cfg in "3" "3 --xflags 16" "2" "5" "4"; do
    n=$(echo $cfg | tr -d ' -')
    timeout 300 python bench.py --resident off --no-cpu-baseline --config $cfg --steps 400 --warmup 50 --repeats 2 2>/dev/null | grep '"metric"' | python -c '
import sys, json
j = json.loads(sys.stdin.readline()); r = j["roofline"]
print("licm_'$lib' config '"$n"' ms_per_step %.5f repeat %.5f kernel_ms %.5f" % (j["ms_per_step"], j["config"]["repeat_ms_per_step"]["median"], r["kernel_ms"]))' | tee -a "$out/summary.txt"
  done
done
cp /tmp/lib_off.so openal-soft_amd/liboalgpu.so
