export TMPDIR=/tmp
mkdir -p gpurun_out/r2phase
for vpg in 0 16; do VPG=$vpg timeout 200 python tools/block_phase_times.py 2>&1 | tail -12 | tee gpurun_out/r2phase/block4_vpg$vpg.txt; done
