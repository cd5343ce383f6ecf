#!/bin/bash
# PMC passes of lbm_wave<8> / lbm_wave<6> on 8192^2
set -o pipefail
O=gpurun_out/r02h
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp LBM_MARCH_KERNEL=1 LBM_WAVE_ROWS=64
for K in 8 6; do
export LBM_TIME_BLOCK=$K
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $O/sq$K --output-format csv -- python3 bench.py --workload 8192x8192 --steps 48 --warmup 0 --also '' --cpu-sample-steps 0 > $O/sq$K.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f$K --output-format csv -- python3 bench.py --workload 8192x8192 --steps 48 --warmup 0 --also '' --cpu-sample-steps 0 > $O/f$K.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w$K --output-format csv -- python3 bench.py --workload 8192x8192 --steps 48 --warmup 0 --also '' --cpu-sample-steps 0 > $O/w$K.log 2>&1 || exit 1
done
tail -2 $O/sq8.log
