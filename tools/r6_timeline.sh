# measurement aid: the step's timeline under rocprofv3 for config $1 with --xflags $2 (and --resident off): kernel stats + tools/step_timeline.py
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6tl_$2; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python bench.py --config $1 --xflags $2 --resident off --steps 100 --warmup 5 --repeats 0 --no-cpu-baseline < /dev/null > $O/prof.log 2>&1
cp $(find $O/prof -name "p_kernel_stats.csv" | head -1) $O/kernel_stats.csv
python tools/step_timeline.py $(find $O/prof -name "p_kernel_trace.csv" | head -1) 4 0.2 > $O/step_timeline.txt 2>&1
head -8 $O/kernel_stats.csv | cut -c1-60,170-330; cat $O/step_timeline.txt
rm -rf $O/prof
