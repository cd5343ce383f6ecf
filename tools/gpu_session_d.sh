#!/bin/bash
set -o pipefail
O=gpurun_out/r02d
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python tools/march_check.py > $O/march_check.log 2>&1
echo "march_check rc=$?"
cat $O/march_check.log
