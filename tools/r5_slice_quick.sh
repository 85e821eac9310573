# config 4 on the slice kernel: parity cases, one bench line, instruction counters: gpurun -- "bash tools/r5_slice_quick.sh TAG"
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_baseline_configs.py tests/test_delayed_start.py tests/test_queue_adpcm.py tests/test_ambi_voices.py tests/test_gpu_panning.py tests/test_reverb.py -x -q -m gpu -k "not updates_1_2_8_50 and not config3 and not config5 and not config2" 2>&1 | tail -5) > $O/t.log 2>&1
timeout 120 python bench.py --config 4 --xflags ${XF:-128} --steps 200 --warmup 20 --no-cpu-baseline --repeats 1 > $O/bench4.json 2> $O/bench4.err
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$tag -o pmc -- python bench.py --config 4 --xflags ${XF:-128} --steps 6 --warmup 2 --preroll 4 --no-cpu-baseline --repeats 0 < /dev/null > $O/pmc_$tag.log 2>&1
  python - <<PY
import csv,collections,glob
try:
    f=glob.glob("$O/pmc_$tag/**/*counter_collection.csv", recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if "Voice" in r["Kernel_Name"] and "Kernel<" in r["Kernel_Name"]]
    d=collections.defaultdict(list)
    for r in rows: d[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in d.items(): print(k[1], len(v), sorted(v)[len(v)//2])
except Exception as e: print("ERR", "$tag", e)
PY
done > $O/pmc.txt 2>&1
cat $O/t.log
python -c "
import json; d=json.loads(open('$O/bench4.json').read().strip().splitlines()[-1]); print('step', d['ms_per_step'], d['roofline']['kernel'], 'kernel_ms', d['roofline']['kernel_ms'])"
cat $O/pmc.txt
