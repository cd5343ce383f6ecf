"""The row-slab decomposition and its halo protocol, rehearsed on CPU: world_size 2 and 3
over gloo, one slab per rank, the oracle doing each slab's arithmetic.  What this pins is the
PROTOCOL the HIP path implements natively (csrc/lbm_host_slabs.inc, lbm_host_march.inc: which rows and planes travel, the
periodic ring, where the accelerate row lives, how av_vels is reduced); the HIP kernels and
the RCCL transport themselves are covered by the -m gpu tests.

Every halo value that the protocol does NOT send is poisoned with NaN, so a result equal to
the single-domain oracle proves that three planes of one row per direction are sufficient."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, deck_paths


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, deck, nsteps, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import advanced_hpc_lbm_amd as L
    import lbm_oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = O.Oracle("strict")
        pf, of = deck
        gp = O.read_params(pf)
        gob = O.read_obstacles(of, gp.nx, gp.ny)
        nx, ny = gp.nx, gp.ny
        r0, r1 = L.slab_bounds(ny, world, rank)
        nyl = r1 - r0
        south, north = L.ring_neighbours(rank, world)
        # slab + one halo row below and above, as its own little periodic lattice
        lp = O.OrcParam(nx, nyl + 2, gp.maxIters, gp.reynolds_dim, gp.density, gp.accel, gp.omega)
        ob = np.zeros((nyl + 2, nx), np.int32)
        ob[1:nyl + 1] = gob[r0:r1]
        a = np.full((nyl + 2, nx, 9), np.nan)
        a[1:nyl + 1] = orc.init_cells(gp, np.float64)[r0:r1]
        b = np.full_like(a, np.nan)
        accel_local = (ny - 2) - r0 + 1 if r0 <= ny - 2 < r1 else None
        to_s, to_n = list(L.HALO_PLANES_TO_SOUTH), list(L.HALO_PLANES_TO_NORTH)
        av = np.zeros(nsteps)
        for tt in range(nsteps):
            if accel_local is not None:
                orc.accelerate_row(lp, a, ob, accel_local)
            a[0] = np.nan
            a[nyl + 1] = np.nan
            send_s = torch.from_numpy(np.ascontiguousarray(a[1][:, to_s]))
            send_n = torch.from_numpy(np.ascontiguousarray(a[nyl][:, to_n]))
            recv_n = torch.empty_like(send_s)
            recv_s = torch.empty_like(send_n)
            reqs = [dist.isend(send_s, south, tag=1), dist.isend(send_n, north, tag=2),
                    dist.irecv(recv_n, north, tag=1), dist.irecv(recv_s, south, tag=2)]
            for r in reqs:
                r.wait()
            a[nyl + 1][:, to_s] = recv_n.numpy()   # the northern neighbour's row 0, planes 4,7,8
            a[0][:, to_n] = recv_s.numpy()         # the southern neighbour's top row, planes 2,5,6
            tot, cnt = orc.sweep_rows(lp, a, b, ob, 1, nyl + 1)
            red = torch.tensor([tot, float(cnt)], dtype=torch.float64)
            dist.all_reduce(red)
            av[tt] = red[0].item() / red[1].item()
            a, b = b, a
        np.save(os.path.join(outdir, f"state_{rank}.npy"), a[1:nyl + 1])
        np.save(os.path.join(outdir, f"av_{rank}.npy"), av)
    finally:
        dist.destroy_process_group()


def _worker_pairs(rank, world, port, deck, npairs, outdir):
    """Two steps per halo exchange, two halo rows per side holding only the nine slots the
    two-step kernel trades (everything else NaN)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import advanced_hpc_lbm_amd as L
    import lbm_oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = O.Oracle("strict")
        pf, of = deck
        gp = O.read_params(pf)
        gob = O.read_obstacles(of, gp.nx, gp.ny)
        nx, ny = gp.nx, gp.ny
        r0, r1 = L.slab_bounds(ny, world, rank)
        nyl = r1 - r0
        south, north = L.ring_neighbours(rank, world)
        # ext rows: 0 <-> local -2, 1 <-> -1, 2..nyl+1 <-> own rows, nyl+2 <-> nyl, nyl+3 <-> nyl+1
        lp = O.OrcParam(nx, nyl + 4, gp.maxIters, gp.reynolds_dim, gp.density, gp.accel, gp.omega)
        ob = np.zeros((nyl + 4, nx), np.int32)
        ob[2:nyl + 2] = gob[r0:r1]
        ob[1] = gob[(r0 - 1) % ny]          # the ring rows' blocked maps (static, known at create)
        ob[nyl + 2] = gob[r1 % ny]
        a = np.full((nyl + 4, nx, 9), np.nan)
        a[2:nyl + 2] = orc.init_cells(gp, np.float64)[r0:r1]
        b = np.full_like(a, np.nan)

        def ext_row_of_global(g):   # ext index of global row g if it is an own or ring row, else None
            for e in range(1, nyl + 3):
                if (r0 - 2 + e) % ny == g:
                    return e
            return None

        acc_e = ext_row_of_global(ny - 2)
        own_acc = acc_e if acc_e is not None and 2 <= acc_e <= nyl + 1 else None
        av = np.zeros(2 * npairs)
        for j in range(npairs):
            if own_acc is not None:                   # accelerate of the pair's first step (own row only;
                orc.accelerate_row(lp, a, ob, own_acc)  # the neighbours see it through the halo)
            a[0:2] = np.nan
            a[nyl + 2:] = np.nan
            send_s = torch.from_numpy(np.stack([a[2 + dr][:, k] for dr, k in L.HALO9_TO_SOUTH]))
            send_n = torch.from_numpy(np.stack([a[nyl + 1 - dr][:, k] for dr, k in L.HALO9_TO_NORTH]))
            recv_n, recv_s = torch.empty_like(send_s), torch.empty_like(send_n)
            reqs = [dist.isend(send_s, south, tag=1), dist.isend(send_n, north, tag=2),
                    dist.irecv(recv_n, north, tag=1), dist.irecv(recv_s, south, tag=2)]
            for r in reqs:
                r.wait()
            for slot, (dr, k) in enumerate(L.HALO9_TO_SOUTH):   # from the north: its rows 0, 1 = my nyl, nyl+1
                a[nyl + 2 + dr][:, k] = recv_n[slot].numpy()
            for slot, (dr, k) in enumerate(L.HALO9_TO_NORTH):   # from the south: its top rows = my -1, -2
                a[1 - dr][:, k] = recv_s[slot].numpy()
            # first step: own rows (counted) + the two ring rows (recomputed, not counted)
            b[:] = np.nan
            orc.sweep_rows(lp, a, b, ob, 1, 2)
            tot, cnt = orc.sweep_rows(lp, a, b, ob, 2, nyl + 2)
            orc.sweep_rows(lp, a, b, ob, nyl + 2, nyl + 3)
            red = torch.tensor([tot, float(cnt)], dtype=torch.float64)
            dist.all_reduce(red)
            av[2 * j] = red[0].item() / red[1].item()
            if acc_e is not None:                     # accelerate of the second step, ring rows included
                orc.accelerate_row(lp, b, ob, acc_e)
            a[:] = np.nan
            tot, cnt = orc.sweep_rows(lp, b, a, ob, 2, nyl + 2)
            red = torch.tensor([tot, float(cnt)], dtype=torch.float64)
            dist.all_reduce(red)
            av[2 * j + 1] = red[0].item() / red[1].item()
        np.save(os.path.join(outdir, f"state_{rank}.npy"), a[2:nyl + 2])
        np.save(os.path.join(outdir, f"av_{rank}.npy"), av)
    finally:
        dist.destroy_process_group()


def _worker_march(rank, world, port, deck, ngroups, K, outdir):
    """K steps per exchange, the marching kernels' ghost zone: K whole rows of the neighbour either side (the HIP path
    reads them in place out of the neighbour's lattice, here they travel over gloo), recomputed with shrinking reach;
    everything a sub-step does not compute is NaN."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import advanced_hpc_lbm_amd as L
    import lbm_oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = O.Oracle("strict")
        pf, of = deck
        gp = O.read_params(pf)
        gob = O.read_obstacles(of, gp.nx, gp.ny)
        nx, ny = gp.nx, gp.ny
        r0, r1 = L.slab_bounds(ny, world, rank)
        nyl = r1 - r0
        south, north = L.ring_neighbours(rank, world)
        ne = nyl + 2 * K                               # ext row e <-> slab row e - K <-> lattice row (r0 - K + e) mod ny
        lp = O.OrcParam(nx, ne, gp.maxIters, gp.reynolds_dim, gp.density, gp.accel, gp.omega)
        ob = np.stack([gob[(r0 - K + e) % ny] for e in range(ne)]).astype(np.int32)   # neighbours' obstacle rows: static
        a = np.full((ne, nx, 9), np.nan)
        a[K:K + nyl] = orc.init_cells(gp, np.float64)[r0:r1]
        b = np.full_like(a, np.nan)
        acc_rows = [e for e in range(ne) if (r0 - K + e) % ny == ny - 2]   # the accelerate row and its images in the ghost zone
        av = np.zeros(K * ngroups)
        for g in range(ngroups):
            a[:K] = np.nan
            a[K + nyl:] = np.nan
            send_s = torch.from_numpy(np.ascontiguousarray(a[K:2 * K]))              # my bottom K rows -> the south's northern ghost
            send_n = torch.from_numpy(np.ascontiguousarray(a[nyl:nyl + K]))          # my top K rows -> the north's southern ghost
            recv_n, recv_s = torch.empty_like(send_s), torch.empty_like(send_n)
            reqs = [dist.isend(send_s, south, tag=1), dist.isend(send_n, north, tag=2),
                    dist.irecv(recv_n, north, tag=1), dist.irecv(recv_s, south, tag=2)]
            for r in reqs:
                r.wait()
            a[K + nyl:] = recv_n.numpy()
            a[:K] = recv_s.numpy()
            for s_ in range(1, K + 1):                 # sub-step s_ is valid on ext rows [s_, ne - s_)
                for e in acc_rows:
                    if s_ - 1 <= e < ne - (s_ - 1):    # (a row of the current state that is still valid)
                        orc.accelerate_row(lp, a, ob, e)
                b[:] = np.nan
                if s_ < K:
                    orc.sweep_rows(lp, a, b, ob, s_, K)
                    orc.sweep_rows(lp, a, b, ob, K + nyl, ne - s_)
                tot, cnt = orc.sweep_rows(lp, a, b, ob, K, K + nyl)
                red = torch.tensor([tot, float(cnt)], dtype=torch.float64)
                dist.all_reduce(red)
                av[g * K + s_ - 1] = red[0].item() / red[1].item()
                a, b = b, a
        np.save(os.path.join(outdir, f"state_{rank}.npy"), a[K:K + nyl])
        np.save(os.path.join(outdir, f"av_{rank}.npy"), av)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,deck,ngroups,K", [(2, "128x256", 6, 4), (2, "128x128", 6, 4), (3, "128x128", 4, 4), (2, "128x256", 3, 6), (2, "128x256", 3, 8), (3, "128x128", 2, 8)])
def test_k_row_ghost_zone_of_the_marching_kernels(tmp_path, O, oracle, world, deck, ngroups, K):
    """K steps on a slab need K rows of each neighbour and nothing else (accelerate row included, wherever in the ghost
    zone its periodic image falls): what lbm_march reads in place across slabs (csrc/lbm_host_march.inc: launch_march_slabs; lbm_host_run.inc: run_p2p),
    rehearsed with the oracle over gloo, NaN everywhere the algorithm does not compute."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.start_processes(_worker_march, args=(world, port, deck_paths(deck), ngroups, K, str(tmp_path)),
                       nprocs=world, join=True, start_method="spawn")
    pf, of = deck_paths(deck)
    prm = O.read_params(pf)
    ob = O.read_obstacles(of, prm.nx, prm.ny)
    cells = oracle.init_cells(prm, np.float64)
    av = oracle.run(prm, cells, ob, K * ngroups)
    import advanced_hpc_lbm_amd as L
    for r in range(world):
        r0, r1 = L.slab_bounds(prm.ny, world, r)
        got = np.load(tmp_path / f"state_{r}.npy")
        assert np.array_equal(got, cells[r0:r1]), f"slab {r} differs from the single-domain lattice"
        assert np.allclose(np.load(tmp_path / f"av_{r}.npy"), av, rtol=1e-12, atol=0)


@pytest.mark.parametrize("world,deck,npairs", [(2, "128x256", 15), (2, "128x128", 15), (3, "128x128", 8)])
def test_two_step_halo_protocol_matches_single_domain_oracle(tmp_path, O, oracle, world, deck, npairs):
    """The nine-slot halo of the two-step kernel (HALO9_*): exchanged once per pair of steps, ring rows
    recomputed on both sides -- sufficient (all else is NaN) and bit-exact against the single domain."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.start_processes(_worker_pairs, args=(world, port, deck_paths(deck), npairs, str(tmp_path)),
                       nprocs=world, join=True, start_method="spawn")
    pf, of = deck_paths(deck)
    prm = O.read_params(pf)
    ob = O.read_obstacles(of, prm.nx, prm.ny)
    cells = oracle.init_cells(prm, np.float64)
    av = oracle.run(prm, cells, ob, 2 * npairs)
    import advanced_hpc_lbm_amd as L
    for r in range(world):
        r0, r1 = L.slab_bounds(prm.ny, world, r)
        got = np.load(tmp_path / f"state_{r}.npy")
        assert np.array_equal(got, cells[r0:r1]), f"slab {r} differs from the single-domain lattice"
        assert np.allclose(np.load(tmp_path / f"av_{r}.npy"), av, rtol=1e-12, atol=0)


@pytest.mark.parametrize("world,deck,nsteps", [(2, "128x256", 40), (2, "128x128", 40), (3, "128x128", 25)])
def test_slab_protocol_matches_single_domain_oracle(tmp_path, O, oracle, world, deck, nsteps):
    """One step per exchange: each rank sends planes 4,7,8 of its row 0 south and planes 2,5,6 of its top row north
    (d2q9-bgk.c:971-998), everything else in the halo rows is NaN.  This is the message content of the one-step halo
    kernels AND of the register tiles across slabs (csrc/lbm_regtile.hip.h, kRegSlab: the same three populations per column
    and step, as 16-byte granules stored into the neighbour's mailboxes) -- only the transport differs."""
    import torch.multiprocessing as mp
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, deck_paths(deck), nsteps, str(tmp_path)),
                       nprocs=world, join=True, start_method="spawn")
    pf, of = deck_paths(deck)
    prm = O.read_params(pf)
    ob = O.read_obstacles(of, prm.nx, prm.ny)
    cells = oracle.init_cells(prm, np.float64)
    av = oracle.run(prm, cells, ob, nsteps)
    import advanced_hpc_lbm_amd as L
    for r in range(world):
        r0, r1 = L.slab_bounds(prm.ny, world, r)
        got = np.load(tmp_path / f"state_{r}.npy")
        assert np.array_equal(got, cells[r0:r1]), f"slab {r} differs from the single-domain lattice"
        assert np.allclose(np.load(tmp_path / f"av_{r}.npy"), av, rtol=1e-12, atol=0)


def test_partition_covers_every_row_once(L):
    for ny in (2, 7, 128, 1024, 8192):
        for n in (1, 2, 3, 4, 8):
            if n > ny:
                continue
            edges = [L.slab_bounds(ny, n, r) for r in range(n)]
            assert edges[0][0] == 0 and edges[-1][1] == ny
            assert all(edges[i][1] == edges[i + 1][0] for i in range(n - 1))
            assert all(e[1] > e[0] for e in edges)
    assert L.ring_neighbours(0, 4) == (3, 1) and L.ring_neighbours(3, 4) == (2, 0)
    assert L.ring_neighbours(0, 1) == (0, 0)
