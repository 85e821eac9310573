cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputest_r4_final.log 2>&1; tail -5 gpurun_out/gputest_r4_final.log
bash tools/r4_evidence.sh
