#!/bin/bash
set -o pipefail
O=gpurun_out/r02f
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 ./tools/layout_bench 8192 241 5 > $O/layout_bench.log 2>&1; echo "layout rc=$?"; cat $O/layout_bench.log
timeout -k 10 300 ./tools/layout_bench 8192 304 5 >> $O/layout_bench.log 2>&1
timeout -k 10 600 python tools/march_check.py > $O/march_check.log 2>&1
echo "march_check rc=$?"
cat $O/march_check.log
