timeout 900 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | tail -2
for c in 3 2 4 5; do echo -n "config $c: "; timeout 300 python bench.py --config $c --steps 200 --warmup 20 --no-cpu-baseline < /dev/null 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'])"; done
