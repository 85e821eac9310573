export TMPDIR=/tmp
O=gpurun_out/r2occ
mkdir -p $O
for w in 4 3; do
OALGPU_REPORT_OCCUPANCY=1 OALGPU_BLOCK_WAVES=$w timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 < /dev/null > $O/bench_$w.json 2> $O/bench_$w.err
grep -h "workgroups per CU" $O/bench_$w.err
python -c "import json;d=json.load(open('$O/bench_$w.json'));print($w, round(d['value']/1e6,2),'Mv/s step', round(d['ms_per_step']*1e3,2),'us kernel', round(d['roofline']['kernel_ms']*1e3,2))"
for vpg in 1 2 8 16; do
OALGPU_REPORT_OCCUPANCY=1 OALGPU_BLOCK_WAVES=$w timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --vpg $vpg < /dev/null > $O/bench_${w}_$vpg.json 2> $O/bench_${w}_$vpg.err
grep -h "workgroups per CU" $O/bench_${w}_$vpg.err
python -c "import json;d=json.load(open('$O/bench_${w}_$vpg.json'));print($w, 'vpg',$vpg, round(d['value']/1e6,2),'Mv/s step', round(d['ms_per_step']*1e3,2),'us kernel', round(d['roofline']['kernel_ms']*1e3,2))"
done
done
