export TMPDIR=/tmp
mkdir -p gpurun_out/r1n
for c in 3 2 4 5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1n/prof_c$c -o p -- python bench.py --config $c --steps 60 --warmup 5 --no-cpu-baseline < /dev/null > gpurun_out/r1n/prof_c$c.log 2>&1
  f=$(ls gpurun_out/r1n/prof_c$c/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then echo "== config $c"; head -8 "$f" | cut -c1-190; fi
done
