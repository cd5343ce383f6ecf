#!/bin/bash
# Round-2 GPU session A: test suite, driver-style bench, SQ counters of lbm_sweep2 on 8192^2.
set -o pipefail
O=gpurun_out/r02a
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
echo "== pytest" && timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
echo "== bench driver-style" && timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err && cat $O/bench_driver.json &&
echo "== bench default" && timeout -k 10 300 python bench.py --cpu-sample-steps 0 > $O/bench_default.json 2> $O/bench_default.err && cat $O/bench_default.json &&
rocprofv3 -L > $O/counters.txt 2>&1
echo "== pmc A" && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $O/pmcA --output-format csv -- python3 bench.py --workload 8192x8192 --steps 40 --warmup 4 --also '' --cpu-sample-steps 0 > $O/pmcA.log 2>&1 &&
echo "== pmc B" && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/pmcB --output-format csv -- python3 bench.py --workload 8192x8192 --steps 40 --warmup 4 --also '' --cpu-sample-steps 0 > $O/pmcB.log 2>&1
echo "pmc rc=$?"
ls -R $O | head -50
