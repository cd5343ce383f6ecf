#!/bin/bash
# One gpurun call that produces the round's measurement evidence from ONE tree on ONE MI355X:
#   the GPU test suite, the counter passes (tools/profile_run.py under rocprofv3 --pmc, one pass per counter group),
#   rocprofv3 --kernel-trace --stats of the default bench run, the plain bench line and the driver's form.
# Afterwards (here, not on the GPU box):  python tools/collect_counters.py <tag> "<note>" gpurun_out/<tag>/pmc_*
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/evidence_session.sh r03final'
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-evidence}; mkdir -p $O
[ "${2:-}" = "nopmc" ] || { timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log; }
[ "${2:-}" = "nopmc" ] || for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
  d=$O/pmc_$(echo $grp | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 tools/profile_run.py > $d.log 2>&1 || echo "pass $grp failed"
done
# the counters of THIS tree are what the bench lines below are priced with (bench.py reads profiles/kernel_counters.json and
# profiles/hbm_traffic.json); the three files also come back under gpurun_out/ for the commit
if [ "${2:-}" != "nopmc" ]; then
  python tools/collect_counters.py r03 "gpurun session ${1:-evidence} (one MI355X), the tree of that call" $(ls -d $O/pmc_*/) > $O/collect.log 2>&1 || echo "collect_counters failed"
  cp profiles/kernel_counters.json profiles/hbm_traffic.json profiles/r03_sq_counters.json $O/
fi
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 bench.py --warmup 0 --cpu-sample-steps 0 > $O/bench_profiled.json 2> $O/bench_profiled.err; tail -c 300 $O/bench_profiled.json
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; head -c 300 $O/bench_driver.json
ls $O
