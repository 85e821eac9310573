# measurement aid: A/B of two builds of the library in ONE gpurun call -- the tree's liboalgpu.so against openal-soft_amd/liboalgpu_exp.so
# (an experimental build placed there by hand); bench line of config $1 (default 3) for each, twice, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
C=${1:-3}
cp openal-soft_amd/liboalgpu.so /tmp/lib_default.so
for r in 1 2; do
  for v in default exp; do
    if [ $v = exp ]; then cp openal-soft_amd/liboalgpu_exp.so openal-soft_amd/liboalgpu.so; else cp /tmp/lib_default.so openal-soft_amd/liboalgpu.so; fi
    timeout 300 python bench.py --config $C --no-cpu-baseline --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', 'step %.2f us' % (d['ms_per_step']*1e3), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3))"
  done
done
cp /tmp/lib_default.so openal-soft_amd/liboalgpu.so
if [ "$2" = "test" ]; then cp openal-soft_amd/liboalgpu_exp.so openal-soft_amd/liboalgpu.so; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernel_variants.py -x -q 2>&1 | tail -2; fi
