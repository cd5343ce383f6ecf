set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "reference_form or ghost_bands or rccl" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
tools/oob_store_order > $O/oob_store_order.log 2>&1; tail -8 $O/oob_store_order.log
python tools/strong_scaling_proxy.py 8192 "p2p default,rccl" > $O/proxy_default.log 2>&1; grep us/step $O/proxy_default.log
LBM_TIME_BLOCK=6 LBM_WAVE_COLS=2 python tools/strong_scaling_proxy.py 8192 "p2p default,rccl" > $O/proxy_k6x2.log 2>&1; grep us/step $O/proxy_k6x2.log
LBM_TIME_BLOCK=8 LBM_WAVE_COLS=2 python tools/strong_scaling_proxy.py 8192 "p2p default,rccl" > $O/proxy_k8x2.log 2>&1; grep us/step $O/proxy_k8x2.log
LBM_BAND_EDGE_ROWS=12 python tools/strong_scaling_proxy.py 8192 "rccl" > $O/proxy_e12.log 2>&1; grep us/step $O/proxy_e12.log
LBM_BAND_EDGE_ROWS=30 python tools/strong_scaling_proxy.py 8192 "rccl" > $O/proxy_e30.log 2>&1; grep us/step $O/proxy_e30.log
python tools/wave_sweep.py 8192 --k 8 --cols 1 --rows 128 156 160 240 256 --no-march > $O/sweep_rows_k8.log 2>&1; cat $O/sweep_rows_k8.log
python tools/wave_sweep.py 8192 --k 6 --cols 2 --rows 56 64 76 --no-march > $O/sweep_rows_k6x2.log 2>&1; cat $O/sweep_rows_k6x2.log
