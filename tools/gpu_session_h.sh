#!/bin/bash
# regtile timing session: default + debug variants + poll statistics
O=gpurun_out/r02m; mkdir -p $O
{
for d in 0 1 2 3 4 5; do LBM_RESIDENT_DEBUG=$d timeout -k 10 120 python tools/regtile_timing.py 1024x1024 2000 644; done
LBM_REGTILE_STATS=1 timeout -k 10 120 python tools/regtile_timing.py 1024x1024 2000 644
LBM_REGTILE_STATS=1 timeout -k 10 120 python tools/regtile_timing.py 256x256 2000 41 82
LBM_REGTILE_STATS=1 timeout -k 10 120 python tools/regtile_timing.py 128x128 2000 21
} > $O/regtile_timing.log 2>&1
cat $O/regtile_timing.log
