# quick iteration: correctness + phase times + bench of the workgroup-per-voice kernel
export TMPDIR=/tmp
O=gpurun_out/r2iter
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 300 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "config3 and synthetic" 2>&1 | grep "dry/real\|accumulator" | head -12
OALGPU_VOICE_KERNEL=wave OALGPU_FIR=valu timeout 300 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q -s -k "config3 and synthetic" 2>&1 | grep "dry/real\|accumulator\|passed\|failed" | head -12
VPG=0 timeout 200 python tools/block_phase_times.py 2>&1 | tail -9 | tee $O/phase.txt
for w in 4 3; do
OALGPU_BLOCK_WAVES=$w timeout 300 python bench.py --no-cpu-baseline --steps 500 --warmup 50 < /dev/null > $O/bench_$w.json 2> $O/bench_$w.err
python -c "import json;d=json.load(open('$O/bench_$w.json'));print('block$w', round(d['value']/1e6,2),'Mv/s step', round(d['ms_per_step']*1e3,2),'us kernel', round(d['roofline']['kernel_ms']*1e3,2))"
done
