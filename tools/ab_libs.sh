# measurement aid: bench lines of several builds of the library in ONE gpurun call, twice each, interleaved.
#   bash tools/ab_libs.sh CONFIG XFLAGS STEPS name1=path1.so name2=path2.so ...     (paths relative to the repo; the tree's liboalgpu.so is "tree")
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
C=$1; X=$2; S=$3; shift 3
cp openal-soft_amd/liboalgpu.so /tmp/lib_tree.so
for r in 1 2; do
  for nv in tree=/tmp/lib_tree.so "$@"; do
    n=${nv%%=*}; p=${nv#*=}
    cp $p openal-soft_amd/liboalgpu.so
    timeout 300 python bench.py --config $C --xflags $X --steps $S --warmup 20 --no-cpu-baseline --repeats 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-14s' % '$n', 'step %.2f us' % (d['ms_per_step']*1e3), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3), d['roofline']['kernel'][:24])"
  done
done
cp /tmp/lib_tree.so openal-soft_amd/liboalgpu.so
