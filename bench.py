#!/usr/bin/env python3
"""Headline benchmark: MLUPS of the D2Q9-BGK time-step path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 1024x1024|8192x8192]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one lattice sweep (the reference's timestep_new2 + swap, d2q9-bgk.c:182,190)
of the whole lattice.  The default workload is the configuration BASELINE.json quotes its
metric on -- the shipped 1024x1024 deck -- row-partitioned over the N GPUs (strong scaling);
the synthetic 8192x8192 lattice of BASELINE.json is measured in the same run and reported
under "also".  Inputs are resident in HBM before the timed region; exactly K steps are
timed between barrier + torch.cuda.synchronize() pairs; the maximum over ranks is used.

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel against TWO roofs and names the
one that binds it (`bound`: "hbm", "valu" or "latency"):
  hbm   bytes that crossed the HBM interface per launch (rocprofv3 PMC passes, profiles/hbm_traffic.json;
        where there is no pass for the kernel: the minimum its blocking scheme must move, kernel_of)
        / the mean launch time measured live with HIP events on the library's compute stream / 8 TB/s;
  valu  vector instructions issued per launch (SQ_INSTS_VALU, profiles/kernel_counters.json) / the same
        launch time / the chip's issue rate (256 CUs x 4 SIMDs x one wave-instruction per 2 cycles x 2.4 GHz);
`frac` is the larger of the two.  The temporally blocked kernels move far fewer than the one-step
kernel's 72 B per update, so the rate priced at 72 B is reported beside it as
equiv_72B_* and may exceed 1.  `cpu_baseline` times the serial reference on this host
(oracle/_ref/d2q9-bgk, built from the reference's own source; else our port) on a bounded
sample of the same deck.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before the HIP library: both link libamdhip64.so.7)
import torch.distributed as dist  # noqa: E402

import advanced_hpc_lbm_amd as L  # noqa: E402

MULTI = False            # set in main(): more than one rank, or --rehearse-multi on one GPU
BYTES_PER_LUP = 72.0     # 9 float32 reads + 9 float32 writes
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
# vector issue roof: 256 CUs x 4 SIMDs, a SIMD issues one wave64 VALU instruction per 2 cycles (SIMD-32), 2.4 GHz
VALU_PEAK_WAVE_INSTS = 256 * 4 * 0.5 * 2.4e9
FP32_PEAK_TFLOPS = 157.3  # = that rate x 64 lanes x 2 flop (every slot an FMA), same guide
# what a stream of nothing but independent register-only v_fmac / v_add / v_mul sustains on this part, measured
# (tools/valu_forms, profiles/r03_valu_issue_forms.log: 0.86-0.93e9 per second and SIMD at 2 or 4 waves per SIMD): reported
# beside the spec-based fraction, never instead of it
VALU_MEASURED_WAVE_INSTS = 256 * 4 * 0.90e9


sys.path.insert(0, os.path.join(ROOT, "tools"))
from make_deck import obstacle_map, wall_x  # noqa: E402  (the generator of the synthetic decks the CLI reads)


def synthetic_obstacles(n: int) -> np.ndarray:
    """BASELINE.json's synthetic deck: the 1024x1024 geometry scaled by n/1024 -- closed box
    plus a full-height wall at x = n/3 (2730 for 8192); tools/make_deck.py writes the same map
    as the params + obstacle files the CLI reads."""
    return obstacle_map(n, n)


def make_workload(name: str):
    if name == "1024x1024":
        p = L.read_params(os.path.join(ROOT, "input_1024x1024.params"))
        ob = L.read_obstacles(os.path.join(ROOT, "obstacles_1024x1024.dat"), p)
        data = "shipped deck input_1024x1024.params + obstacles_1024x1024.dat, rest-equilibrium start"
    else:
        m = re.fullmatch(r"(\d+)x(\d+)", name)
        if not m or m.group(1) != m.group(2):
            raise SystemExit(f"unknown workload {name}")
        n = int(m.group(1))
        p = L.Param(n, n, 1000, 10, 0.1, 0.01, 1.85)
        ob = synthetic_obstacles(n)
        data = f"synthetic {n}x{n}: box walls + full-height wall at x={wall_x(n)}, rest-equilibrium start"
    return p, ob, data


def new_unique_id(world, rank):
    """A fresh ncclUniqueId from rank 0, broadcast to every rank (one per lattice = one per communicator)."""
    if not MULTI:
        return None
    buf = torch.zeros(128, dtype=torch.uint8, device="cuda")
    if rank == 0:
        buf.copy_(torch.frombuffer(bytearray(L.rccl_unique_id()), dtype=torch.uint8))
    dist.broadcast(buf, src=0)
    return bytes(buf.cpu().numpy().tobytes())


def make_lattice(p, ob, world, rank, local_rank, exchange):
    if not MULTI:
        return L.Lattice(p, ob, nslabs=1, devices=[local_rank])
    return L.Lattice(p, ob, rank=rank, nranks=world, device=local_rank,
                     unique_id=new_unique_id(world, rank), exchange=exchange)


def all_ranks_agree(ok: bool, world) -> bool:
    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device="cuda")
    dist.all_reduce(t)
    return int(t.item()) == 0


def choose_exchange(p, ob, world, rank, local_rank):
    """N > 1: peer-to-peer halos (kernels store into the neighbour's halo block over xGMI, in-kernel
    flags) if -- on THIS machine, now -- a short run is bit-identical to the same run with RCCL
    send/recv halos on every rank; otherwise RCCL.  Returns (mode, note)."""
    if not MULTI:
        return L.EXCHANGE_AUTO, "none (periodic self-wrap)"
    if os.environ.get("LBM_BENCH_EXCHANGE", "") == "rccl":
        return L.EXCHANGE_RCCL, "RCCL send/recv (forced by LBM_BENCH_EXCHANGE)"
    # ground truth for the self-check: the same 22 steps on the WHOLE lattice, alone on this rank's GPU
    # (every decomposition and transport is bit-identical to it by construction and by test)
    with L.Lattice(p, ob, nslabs=1, devices=[local_rank]) as whole:
        av_true = np.concatenate([whole.run(19), whole.run(3)])
        rb, re_ = L.slab_bounds(p.ny, world, rank)
        st_true = whole.read_state()[rb:re_].copy()
    probe_steps = 400 if p.nx * p.ny <= (1 << 22) else 40

    def probe(mode):
        """(ok, note, seconds for probe_steps).  Every rank walks through the same collectives whatever
        happens to it: library calls sit in try blocks, the agreement all-reduces between them."""
        ok, note, secs, lat = True, "", None, None
        try:
            lat = make_lattice(p, ob, world, rank, local_rank, mode)
            if mode == L.EXCHANGE_P2P and int(lat.info("exchange")) != L.EXCHANGE_P2P:
                ok, note = False, "no peer mapping on some rank (library fell back)"
        except Exception as e:
            ok, note = False, f"{type(e).__name__}: {e}"
        ok = all_ranks_agree(ok, world)
        if ok:
            try:
                av = np.concatenate([lat.run(19), lat.run(3)])  # marching groups (8 or 4 steps), a pair, a single; then an odd run
                st = lat.read_state()
                if not (np.array_equal(st.view(np.uint32), st_true.view(np.uint32)) and
                        np.allclose(av, av_true, rtol=2e-6, atol=0)):
                    ok, note = False, "result differs from the undivided lattice"
            except Exception as e:
                ok, note = False, f"{type(e).__name__}: {e}"
            ok = all_ranks_agree(ok, world)
        if ok:
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            try:
                lat.run(probe_steps)                            # which transport is faster HERE
            except Exception as e:
                ok, note = False, f"{type(e).__name__}: {e}"
            dist.barrier()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            secs = t.item()
            ok = all_ranks_agree(ok, world)
        dist.barrier()
        try:
            if lat is not None:
                lat.close()
        except Exception:
            pass
        return ok, note, secs

    ok_r, note_r, secs_r = probe(L.EXCHANGE_RCCL)
    extra = ""
    if not ok_r:
        # no quiet fallback to one step per launch with one-row RCCL halos: that path costs ~200 us per
        # step on 1024^2 (profiles/r01_slab_overheads_one_gpu.log), which is not a benchmark of anything
        raise SystemExit(f"RCCL halo exchange failed its check against the undivided lattice: {note_r}")
    ok_p, note_p, secs_p = probe(L.EXCHANGE_P2P)
    if not ok_p:
        return L.EXCHANGE_RCCL, "RCCL send/recv (peer-to-peer halos not usable here: %s)%s" % (note_p or "failure on another rank", extra)
    if secs_p > 1.05 * secs_r:
        return L.EXCHANGE_RCCL, ("RCCL send/recv (peer-to-peer halos passed the check against the undivided lattice but were "
                                 "slower here: %.1f vs %.1f ms per %d steps)%s" % (secs_p * 1e3, secs_r * 1e3, probe_steps, extra))
    return L.EXCHANGE_P2P, ("peer-to-peer: edge tiles store their halo rows (9*nx floats per direction per pair of steps) straight "
                            "into the neighbour's halo block over xGMI, in-kernel flags; checked at start-up on this machine: "
                            "bit-identical to the undivided lattice, as is RCCL send/recv; %.1f vs %.1f ms per %d steps%s"
                            % (secs_p * 1e3, secs_r * 1e3, probe_steps, extra))


def kernel_of(lat, tb):
    """(name, minimum HBM bytes per lattice update) of the kernel the timed run used.  The minimum is what
    the kernel's own blocking scheme must move (obstacle bytes excluded):
      lbm_sweep    one step per pass                      36 read + 36 written          = 72
      lbm_sweep2   two steps per pass, 64x16 tiles + ring  (36 x 66x18/(64x16) + 36) / 2 = 38.9
      lbm_march    four steps per pass, 224 of 256 columns (36 x 258/224 + 36) / 4       = 19.4
      lbm_wave<K>  K steps per pass, 64-2K of 64 columns   (36 x 64/(64-2K) + 36) / K    = 19.3 / 13.7 / 10.5 (K = 4 / 6 / 8)
      lbm_wave<8>x2  two columns per lane, 112 of 128      (36 x 128/112 + 36) / 8       = 9.64
      lbm_regtile  the whole run on chip (registers): one load and one store of the lattice per RUN; the figure
                   reported is that, per step -- the kernel is bound by arithmetic and by the hand-off between
                   neighbouring tiles, not by HBM"""
    if int(lat.info("engine_last")) == 3:
        return "lbm_regtile", None
    if tb >= 4 and int(lat.info("march_kernel")) == 1:       # one wave per 64 (128) columns, 64 - 2K (128 - 2K) delivered
        cols = int(lat.info("wave_cols_active"))
        w = 64 * cols
        return f"lbm_wave<{tb}>" + ("x2" if cols == 2 else ""), (36.0 * w / (w - 2 * tb) + 36.0) / tb
    if tb == 4:
        return "lbm_march<4>", (36.0 * 258 / 224 + 36.0) / 4
    if tb == 2:
        return "lbm_sweep2<64,16>", (36.0 * (66 * 18) / (64 * 16) + 36.0) / 2
    return f"lbm_sweep<{int(lat.info('vector_width'))}>", 72.0


def measure(name, world, rank, local_rank, steps, warmup):
    """Creates the resident lattice, warms up, times exactly `steps` steps; at N > 1 every rank then
    repeats the same steps on the UNDIVIDED lattice alone on its GPU and compares its slab bit for bit."""
    p, ob, data = make_workload(name)
    exchange, halo_note = choose_exchange(p, ob, world, rank, local_rank)
    lat = make_lattice(p, ob, world, rank, local_rank, exchange)
    r0, r1 = lat.slab_rows(0)
    mass0 = lat.total_density()          # global (all-reduced) in every mode
    # W warm-up steps in up to three calls: the first (W - 2 steps) brings up the kernels the timed call will use,
    # the two one-step calls after it take the Python / ctypes / HIP-runtime path of lbm_run twice more -- a cold
    # first call costs ~20 us of host time (tools/run_overhead.py), a sixth of a 20-step window of 1024^2
    for n in ([warmup - 2, 1, 1] if warmup >= 3 else [1] * warmup):
        lat.run(n)

    def fence():
        if MULTI:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    av = lat.run(steps)
    t_run = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    gpu_ms, lib_ms = lat.last_run_ms()
    if MULTI:
        t = torch.tensor([dt, gpu_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, gpu_ms = t[0].item(), t[1].item()
    tb = int(lat.info("time_block_active"))              # steps per launch: 4 lbm_march, 2 lbm_sweep2, 1 lbm_sweep
    kernel, min_bytes = kernel_of(lat, tb)
    # stale or missing halos would break the mass balance at the slab boundaries long before anything
    # goes non-finite; float32 rounding alone drifts ~2e-8 per step (tests/test_gpu_parity.py)
    mass_drift = abs(lat.total_density() - mass0) / mass0
    bitexact = None
    if not MULTI:
        # N = 1: replay warm-up + steps with the ONE-STEP kernel (lbm_sweep: the kernel pinned to the reference's own
        # known answers) on a second lattice and compare the whole state bit for bit, av_vels to summation order
        try:
            mine = lat.read_state()
            with L.Lattice(p, ob, nslabs=1, devices=[local_rank]) as plain:
                plain.set_option("time_block", 1)
                av_w = np.concatenate([plain.run(warmup), plain.run(steps)]) if warmup > 0 else plain.run(steps)
                assert int(plain.info("engine_last")) == 1 and int(plain.info("time_block_active")) == 1
                ref = plain.read_state()
            bitexact = bool(np.array_equal(mine.view(np.uint32), ref.view(np.uint32)) and
                            np.allclose(av, av_w[warmup:], rtol=2e-6, atol=0))
            del mine, ref
        except Exception as e:
            bitexact = False
            print(f"post-run check against the one-step kernel failed: {type(e).__name__}: {e}", file=sys.stderr)
    if MULTI:
        # every decomposition and transport is bit-identical to the undivided lattice by construction
        # and by test: check it on THIS machine for THIS run, all warmup + steps steps of it
        ok = True
        try:
            mine = lat.read_state()
            with L.Lattice(p, ob, nslabs=1, devices=[local_rank]) as whole:
                av_w = np.concatenate([whole.run(warmup), whole.run(steps)]) if warmup > 0 else whole.run(steps)
                ref = whole.read_state()[r0:r1]
            ok = bool(np.array_equal(mine.view(np.uint32), ref.view(np.uint32)) and
                      np.allclose(av, av_w[warmup:], rtol=2e-6, atol=0))
        except Exception as e:          # (every rank still reaches the collective below)
            ok = False
            print(f"rank {rank}: post-run check failed: {type(e).__name__}: {e}", file=sys.stderr)
        bitexact = all_ranks_agree(ok, world)
    fence()
    lat.close()
    cells = p.nx * p.ny
    local_cells = p.nx * (r1 - r0)
    launches = steps // tb + (steps % tb) // 2 + (steps % tb) % 2 if tb >= 4 else steps // tb + steps % tb
    if kernel == "lbm_regtile":
        launches, tb = 1, steps                              # the whole run is one launch
    launch_s = gpu_ms * 1e-3 / launches                  # mean duration of one launch on this GPU
    lups_per_launch = local_cells * steps / launches
    if min_bytes is None:                                # resident: the lattice crosses HBM twice per run
        min_bytes = 72.0 / steps
    achieved = min_bytes * lups_per_launch / launch_s / 1e9
    equiv72 = BYTES_PER_LUP * lups_per_launch / launch_s / 1e9
    traffic = lookup_traffic(name, world, kernel, steps)
    valu = lookup_valu(name, world, kernel, steps)
    # the two roofs (module docstring): bytes that crossed the HBM interface, vector instructions issued
    hbm_bytes = traffic if traffic is not None else min_bytes * lups_per_launch
    hbm_frac = hbm_bytes / launch_s / 1e9 / HBM_PEAK_GBS
    valu_frac = None if valu is None else valu / launch_s / VALU_PEAK_WAVE_INSTS
    family = kernel.split("<")[0]
    if family == "lbm_regtile":
        bound = "latency"        # neither roof: tile-to-tile hand-offs (BINDS)
    elif valu_frac is not None and valu_frac > hbm_frac:
        bound = "valu"
    else:
        bound = "hbm"
    if valu_frac is not None and valu_frac >= hbm_frac:
        top = {"achieved": round(valu_frac * FP32_PEAK_TFLOPS, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": round(valu_frac, 4),
               "achieved_is": "vector issue slots used, priced as FMAs (64 lanes x 2 flop per wave-instruction): SQ_INSTS_VALU per "
                              "launch (profiles/kernel_counters.json) / this run's launch time"}
    else:
        top = {"achieved": round(hbm_bytes / launch_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4),
               "achieved_is": "PMC bytes per launch (profiles/hbm_traffic.json) / this run's launch time" if traffic is not None
                              else "the kernel's minimum bytes per update x updates per launch / this run's launch time"}
    return {
        "wall_us": {"timed_region": round(dt * 1e6, 1), "lbm_run_call": round(t_run * 1e6, 1),
                    "inside_library": round(lib_ms * 1e3, 1), "gpu_events": round(gpu_ms * 1e3, 1)},
        "mlups": cells * steps / dt / 1e6,
        "ms_per_step": dt * 1e3 / steps,
        "gpu_ms_per_step": gpu_ms / steps,
        "roofline": {"bound": bound, **top, "traffic": traffic,
                     "kernel": kernel, "steps_per_launch": tb, "what_binds_it": BINDS.get(family, ""),
                     "hbm_frac": round(hbm_frac, 4),
                     "valu_frac": None if valu_frac is None else round(valu_frac, 4),
                     "valu_frac_of_measured_issue_rate": None if valu is None else round(valu / launch_s / VALU_MEASURED_WAVE_INSTS, 4),
                     "valu_wave_insts_per_launch": valu,
                     "valu_lane_insts_per_lattice_update": None if valu is None else round(valu * 64 / lups_per_launch, 1),
                     # the model figure: the bytes THIS kernel's blocking must move per update (kernel_of) at this
                     # run's rate; the same rate priced at the one-step kernel's 72 B per update is
                     # reported separately and may exceed 1
                     "min_bytes_per_lattice_update": round(min_bytes, 3),
                     "min_bytes_gbs": round(achieved, 1), "min_bytes_frac": round(achieved / HBM_PEAK_GBS, 4),
                     "lattice_updates_per_launch": lups_per_launch,
                     "launch_us": round(launch_s * 1e6, 3),
                     "equiv_72B_gbs": round(equiv72, 1), "equiv_72B_frac": round(equiv72 / HBM_PEAK_GBS, 4),
                     # what actually crossed the HBM interface: PMC bytes per launch (profiles/hbm_traffic.json,
                     # collected on the box and commit named there) / this run's launch time
                     "traffic_gbs": None if traffic is None else round(traffic / launch_s / 1e9, 1),
                     "traffic_frac_of_peak": None if traffic is None else round(traffic / launch_s / 1e9 / HBM_PEAK_GBS, 4)},
        "data": data, "params": p, "blocked": int(ob.sum()), "av_last": float(av[-1]), "finite": bool(np.isfinite(av).all()),
        "halo": halo_note, "mass_drift": mass_drift, "bitexact": bitexact,
    }


# what the committed profiles say limits each kernel: the HBM fraction alone does not tell it
BINDS = {
    "lbm_regtile": "tile-to-tile hand-off latency: the lattice stays in registers (HBM is crossed twice per RUN), a step is a chain "
                   "of sc1 store -> sc1 load hand-offs between neighbouring tiles overlapped with the rows' arithmetic; neither "
                   "roof binds it (valu_frac and hbm_frac both reported)",
    "lbm_wave": "vector issue: its arithmetic, fill rows and undelivered lanes included (valu_lane_insts_per_lattice_update); "
                "HBM traffic is ~10 B per update",
    "lbm_march": "HBM: 19.3 B per update measured, ~0.64 of peak; the 18-stream access pattern alone tops out at ~5.9 TB/s nominal",
    "lbm_sweep2": "HBM: 38.9 B per update",
    "lbm_sweep": "HBM: 72 B per update",
}


def lookup_traffic(name, world, kernel, steps=0):
    """HBM bytes per launch of `kernel` on workload `name` from the committed rocprofv3 PMC passes
    (profiles/hbm_traffic.json, written by tools/summarize_profile.py), or None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    try:
        key = kernel.replace("<", "").replace(">", "") if kernel.startswith("lbm_wave") else kernel.split("<")[0]
        e = json.load(open(path)).get(name, {}).get(key, {})
        if key == "lbm_regtile" and e:          # one launch per run: the lattice once, plus the tiles' mail per step
            return e["hbm_fixed_bytes_per_launch"] + steps * e["hbm_bytes_per_step"]
        return e.get("hbm_bytes_per_launch")
    except Exception:
        return None


def lookup_valu(name, world, kernel, steps=0):
    """Vector wave-instructions per launch of `kernel` on workload `name` (SQ_INSTS_VALU of a committed rocprofv3
    --pmc pass, profiles/kernel_counters.json), or None."""
    path = os.path.join(ROOT, "profiles", "kernel_counters.json")
    if world != 1 or not os.path.exists(path):
        return None
    try:
        key = kernel.replace("<", "").replace(">", "") if kernel.startswith("lbm_wave") else kernel.split("<")[0]
        e = json.load(open(path)).get(name, {}).get(key, {})
        if key == "lbm_regtile" and e:
            return e["valu_wave_insts_per_step"] * steps
        return e.get("valu_wave_insts_per_launch")
    except Exception:
        return None


def cpu_baseline(sample_steps: int):
    """Serial CPU reference on this host, 1 core, on the 1024x1024 deck cut to `sample_steps` steps
    (rate metric; the reference's own 'Elapsed Compute time' interval, d2q9-bgk.c:176-178,204-206)."""
    ncores = os.cpu_count()
    model = ""
    try:
        model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        pass
    exe = os.path.join(ROOT, "oracle", "_ref", "d2q9-bgk")
    lups = 1024 * 1024 * sample_steps
    if os.path.exists(exe):
        with tempfile.TemporaryDirectory() as td:
            pf = os.path.join(td, "sample.params")
            with open(pf, "w") as f:
                f.write(f"1024\n1024\n{sample_steps}\n10\n0.1\n0.01\n1.85\n")
            r = subprocess.run([exe, pf, os.path.join(ROOT, "obstacles_1024x1024.dat")], cwd=td,
                               capture_output=True, text=True)
        m = re.search(r"Elapsed Compute time:\s+([0-9.]+)", r.stdout)
        if r.returncode == 0 and m and float(m.group(1)) > 0:
            secs = float(m.group(1))
            return {"value": round(lups / secs / 1e6, 2), "unit": "MLUPS", "cores": 1, "kind": "reference",
                    "sample": f"reference d2q9-bgk.c (its Makefile flags, -march=x86-64-v3) on the 1024x1024 deck, "
                              f"{sample_steps} of 20000 steps, {secs:.2f} s compute",
                    "host": f"{model}, {ncores} logical cores visible"}
    # no reference binary here: time our own port of it (oracle/, reference flags)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lbm_oracle as O
    orc = O.Oracle("fast")
    prm = O.read_params(os.path.join(ROOT, "input_1024x1024.params"))
    ob = O.read_obstacles(os.path.join(ROOT, "obstacles_1024x1024.dat"), prm.nx, prm.ny)
    cells = orc.init_cells(prm, np.float32)
    t0 = time.perf_counter()
    orc.run(prm, cells, ob, sample_steps)
    secs = time.perf_counter() - t0
    return {"value": round(lups / secs / 1e6, 2), "unit": "MLUPS", "cores": 1, "kind": "port",
            "sample": f"oracle float port (-Ofast) on the 1024x1024 deck, {sample_steps} of 20000 steps, {secs:.2f} s",
            "host": f"{model}, {ncores} logical cores visible"}


def main():
    # stdout carries exactly ONE line, the result: RCCL prints a version banner to fd 1 when a
    # communicator is created, so everything else that writes to fd 1 is sent to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    result_out = os.fdopen(result_fd, "w")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000, help="timed steps of the headline workload")
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="1024x1024")
    ap.add_argument("--also", default="8192x8192", help="second workload reported under 'also' ('' to skip)")
    ap.add_argument("--also-steps", type=int, default=320)
    ap.add_argument("--cpu-sample-steps", type=int, default=1000, help="0 = skip the CPU baseline")
    ap.add_argument("--rehearse-multi", action="store_true",
                    help="one GPU only: run the N>1 code path (RCCL communicator, halo self-check, peer-to-peer "
                         "halos) on a ring of one rank; not a benchmark configuration")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available() or L.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X; the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    global MULTI
    MULTI = world > 1 or args.rehearse_multi
    if MULTI:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:   # rehearsal of the multi-rank code path on one GPU: a ring of one rank
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ["LBM_FORCE_EXCHANGE"] = "1"
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    head = measure(args.workload, world, rank, local_rank, args.steps, args.warmup)
    also = None
    if args.also and args.also != args.workload:
        also = measure(args.also, world, rank, local_rank, args.also_steps, min(args.warmup, 20))

    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample_steps > 0:
        cpu = cpu_baseline(args.cpu_sample_steps)

    if rank == 0:
        p = head["params"]
        line = {
            "metric": "MLUPS (million lattice updates/sec), D2Q9-BGK time-step loop",
            "value": round(head["mlups"], 1),
            "unit": "MLUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(head["ms_per_step"], 6),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": head["data"],
            "config": {"workload": f"d2q9-bgk {args.workload}, {head['blocked']} blocked cells, "
                                   f"density {p.density:g} accel {p.accel:g} omega {p.omega:g}",
                       "nx": p.nx, "ny": p.ny, "decomposition": f"{world} row slab(s), one per GPU" + (" [REHEARSAL of the multi-rank path on a ring of one]" if MULTI and world == 1 else ""),
                       "halo": head["halo"]},
            "roofline": head["roofline"],
            "cpu_baseline": cpu,
            "equiv_72B_frac_whole_job": round(head["mlups"] * BYTES_PER_LUP / 1e3 / (HBM_PEAK_GBS * world), 4),
            "gpu_ms_per_step": round(head["gpu_ms_per_step"], 6),          # HIP events around the step loop
            "value_gpu_events": round(p.nx * p.ny / head["gpu_ms_per_step"] / 1e3, 1),   # MLUPS by that clock
            "speedup_vs_cpu_baseline": None if not cpu else round(head["mlups"] / cpu["value"], 1),
            "wall_us": head["wall_us"],      # the timed region, the lbm_run call in it, the library's own clock, the GPU's
            "results_finite": head["finite"],
            "mass_drift": float("%.3g" % head["mass_drift"]),
            # N = 1: the timed lattice (warm-up + steps) against the one-step kernel's, whole state bit for bit;
            # N > 1: this run's slabs against the undivided lattice on every rank
            "results_bitexact": head["bitexact"],
            "halo_verified": head["bitexact"] if MULTI else None,
            "results_valid": bool(head["finite"] and head["mass_drift"] < 1e-7 * (args.steps + args.warmup) + 1e-6
                                  and head["bitexact"] is not False),
        }
        if also is not None:
            line["also"] = {args.also: {
                "value": round(also["mlups"], 1), "unit": "MLUPS", "steps": args.also_steps,
                "ms_per_step": round(also["ms_per_step"], 6), "roofline": also["roofline"], "data": also["data"],
                "halo": also["halo"], "results_bitexact": also["bitexact"],
                "equiv_72B_frac_whole_job": round(also["mlups"] * BYTES_PER_LUP / 1e3 / (HBM_PEAK_GBS * world), 4)}}
            if also["bitexact"] is False:
                line["results_valid"] = False
        result_out.write(json.dumps(line) + "\n")
        result_out.flush()
    if MULTI:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
