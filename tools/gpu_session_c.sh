#!/bin/bash
set -o pipefail
O=gpurun_out/r02c
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
for d in 0 1 2; do
  export LBM_RESIDENT_DEBUG=$d
  timeout -k 10 120 python tools/resident_timing.py 1024x1024 2000 >> $O/timing.log 2>&1 || exit 1
  timeout -k 10 120 python tools/resident_timing.py 256x256 4000 16x16x1 16x16x4 32x32x4 32x32x1 32x16x2 >> $O/timing.log 2>&1 || exit 1
  timeout -k 10 120 python tools/resident_timing.py 128x128 4000 8x8x1 16x16x1 16x16x4 >> $O/timing.log 2>&1 || exit 1
done
cat $O/timing.log
