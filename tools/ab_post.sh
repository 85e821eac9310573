# measurement aid: like ab_lib.sh, but reports what the post stream's kernels take beside the voice kernel (rocprofv3 kernel stats)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cp openal-soft_amd/liboalgpu.so /tmp/lib_default.so
for v in default exp default exp; do
  if [ $v = exp ]; then cp openal-soft_amd/liboalgpu_exp.so openal-soft_amd/liboalgpu.so; else cp /tmp/lib_default.so openal-soft_amd/liboalgpu.so; fi
  rm -rf /tmp/abp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o p -- python tools/step_period.py 0 3 0 q > /tmp/abp.log 2>&1
  echo "$v: $(grep 'post-process on' /tmp/abp.log)"
  python - <<PY
import csv,glob
f=glob.glob('/tmp/abp/**/p_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'PostFused' in n or 'BusReduceKernel<4>' in n or 'VoiceWave' in n: print('   ', n.split('(')[0][-40:], r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3))
PY
done
cp /tmp/lib_default.so openal-soft_amd/liboalgpu.so
