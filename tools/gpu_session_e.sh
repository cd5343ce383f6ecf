#!/bin/bash
# Round-2 GPU session E: full GPU suite, driver-style + default bench, kernel trace + PMC traffic of the
# default kernels, CLI on the generated 8192^2 deck.
set -o pipefail
O=gpurun_out/r02q
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
echo "== pytest" && timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/pytest.log; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
echo "== bench driver-style" && timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err && cat $O/bench_driver.json &&
echo "== bench default" && timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err && cat $O/bench_default.json &&
echo "== kernel trace" && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 bench.py --warmup 0 --cpu-sample-steps 0 > $O/trace.log 2>&1 &&
echo "== pmc fetch 8192" && timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmcF8 --output-format csv -- python3 bench.py --workload 8192x8192 --steps 40 --warmup 4 --also '' --cpu-sample-steps 0 > $O/pmcF8.log 2>&1 &&
echo "== pmc write 8192" && timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmcW8 --output-format csv -- python3 bench.py --workload 8192x8192 --steps 40 --warmup 4 --also '' --cpu-sample-steps 0 > $O/pmcW8.log 2>&1 &&
echo "== pmc fetch 1024" && timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmcF1 --output-format csv -- python3 bench.py --steps 400 --warmup 20 --also '' --cpu-sample-steps 0 > $O/pmcF1.log 2>&1 &&
echo "== pmc write 1024" && timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmcW1 --output-format csv -- python3 bench.py --steps 400 --warmup 20 --also '' --cpu-sample-steps 0 > $O/pmcW1.log 2>&1 &&
echo "== pmc SQ march" && timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d $O/pmcSQ --output-format csv -- python3 bench.py --workload 8192x8192 --steps 40 --warmup 4 --also '' --cpu-sample-steps 0 > $O/pmcSQ.log 2>&1
echo "pmc rc=$?"
echo "== CLI 8192" && python tools/make_deck.py 8192 8192 --iters 400 --outdir $O/deck > $O/deck.log && (cd $O/deck && LBM_SKIP_FINAL_STATE=1 timeout -k 10 300 ../../../d2q9-bgk input_8192x8192.params obstacles_8192x8192.dat > ../cli_8192.txt 2> ../cli_8192.err; echo "cli rc=$?"; head -3 av_vels.dat; tail -1 av_vels.dat; rm -f obstacles_8192x8192.dat av_vels.dat) ; cat $O/cli_8192.txt
ls $O
