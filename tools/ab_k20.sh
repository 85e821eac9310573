# measurement aid: A/B of two builds of the library in ONE gpurun call -- the tree's liboalgpu.so against openal-soft_amd/liboalgpu_exp.so
# (an experimental build placed there by hand) -- on the driver's command (K = 20): step and voice-kernel time, three times each, interleaved
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
X=${1:-0}
cp openal-soft_amd/liboalgpu.so /tmp/lib_default.so
for r in 1 2 3; do
  for v in default exp; do
    if [ $v = exp ]; then cp openal-soft_amd/liboalgpu_exp.so openal-soft_amd/liboalgpu.so; else cp /tmp/lib_default.so openal-soft_amd/liboalgpu.so; fi
    timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --xflags $X --no-cpu-baseline --repeats 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', 'step %.2f us' % (d['ms_per_step']*1e3), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3), d['roofline']['kernel'][:24], 'repeat median %.2f' % ((d['config'].get('repeat_ms_per_step') or {}).get('median') or 0))"
  done
done
cp /tmp/lib_default.so openal-soft_amd/liboalgpu.so
