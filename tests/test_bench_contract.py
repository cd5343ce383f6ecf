"""bench.py prints exactly one JSON line with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _run(args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_single_gpu(gpu):
    d = _run(["--steps", "400", "--warmup", "20", "--also-steps", "20", "--cpu-sample-steps", "20"])
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 400 and d["warmup"] == 20 and d["higher_is_better"] is True
    assert d["unit"] == "MLUPS" and d["dtype"] == "f32" and d["vs_baseline"] is None and d["scaling"] == "strong"
    assert "workload" in d["config"] and "1024x1024" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 1024 * 1024 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    ro = d["roofline"]
    # two roofs, the binding one named: lbm_regtile is bound by neither (hand-off latency between tiles); its HBM
    # fraction is tiny by design (the lattice crosses HBM twice per run), its vector-issue fraction the larger one
    assert ro["bound"] == "latency" and ro["kernel"] == "lbm_regtile" and ro["steps_per_launch"] == 400
    assert ro["unit"] in ("GB/s", "TFLOP/s") and ro["peak"] == (8000.0 if ro["unit"] == "GB/s" else 157.3)
    assert abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 2e-3 and 0.0 < ro["frac"] <= 1.0
    assert 0.0 < ro["hbm_frac"] <= 1.0 and (ro["valu_frac"] is None or 0.0 < ro["valu_frac"] <= 1.0)
    assert abs(ro["frac"] - max(ro["hbm_frac"], ro["valu_frac"] or 0.0)) < 2e-3
    assert ro["equiv_72B_frac"] > ro["min_bytes_frac"]
    assert abs(ro["equiv_72B_gbs"] / ro["min_bytes_gbs"] * ro["min_bytes_per_lattice_update"] / 72.0 - 1.0) < 0.05
    assert ro["traffic"] is None or ro["traffic"] > 0
    big = d["also"]["8192x8192"]["roofline"]
    assert big["kernel"].startswith("lbm_wave") and big["steps_per_launch"] in (6, 8) and 0.0 < big["frac"] <= 1.0
    assert big["bound"] in ("valu", "hbm") and abs(big["frac"] - max(big["hbm_frac"], big["valu_frac"] or 0.0)) < 2e-3
    # N = 1: the timed lattice was replayed with the one-step kernel and compared bit for bit
    assert d["gpu_ms_per_step"] <= d["ms_per_step"] and d["results_bitexact"] is True and d["results_valid"] is True
    assert d["also"]["8192x8192"]["results_bitexact"] is True
    cb = d["cpu_baseline"]
    assert cb["unit"] == "MLUPS" and cb["cores"] == 1 and cb["kind"] in ("reference", "port") and cb["value"] > 1
    assert d["results_finite"] is True
    assert "8192x8192" in d["also"] and d["also"]["8192x8192"]["value"] > 0


@pytest.mark.gpu
def test_bench_multi_rank_path_rehearsal(gpu):
    """The N > 1 code path (communicator, transport self-check, peer-to-peer halos) on a ring of one rank."""
    d = _run(["--rehearse-multi", "--steps", "400", "--warmup", "20", "--also", "", "--cpu-sample-steps", "0"])
    assert REQUIRED <= set(d)
    assert d["cpu_baseline"] is None
    assert "peer-to-peer" in d["config"]["halo"] or "RCCL" in d["config"]["halo"]
    assert "REHEARSAL" in d["config"]["decomposition"]
    assert d["results_bitexact"] is True and d["halo_verified"] is True and d["results_valid"] is True
