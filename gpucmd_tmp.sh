mkdir -p gpurun_out/r1m
timeout 900 python -m pytest tests -m gpu -x -q < /dev/null > gpurun_out/r1m/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r1m/pytest.log
tail -3 gpurun_out/r1m/pytest.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline < /dev/null > gpurun_out/r1m/bench.json 2> gpurun_out/r1m/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r1m/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
timeout 200 python tools/phase_times.py < /dev/null 2>&1 | tail -22
