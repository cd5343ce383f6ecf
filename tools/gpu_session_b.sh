#!/bin/bash
# Round-2 GPU session B: first light of the resident kernel.
set -o pipefail
O=gpurun_out/r02b
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python tools/resident_check.py > $O/resident_check.log 2>&1
echo "resident_check rc=$?"
cat $O/resident_check.log
