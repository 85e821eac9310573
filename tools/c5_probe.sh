# measurement aid: the step period of configs 5 and 4, pipelined (two streams) against serial (OALGPU_CTX_SERIAL = 4: one stream)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c5probe; mkdir -p $O
for a in "0 4 0" "0 5 0" "0 2 0"; do
  timeout 200 python tools/step_period.py $a 2>&1 | tee -a $O/reduce_lds_period.txt
done
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
