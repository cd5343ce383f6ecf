"""Pins the float oracle bit for bit: (a) against known-answer vectors produced by the
reference's own timestep_new2 (tests/golden/kat_*.npz, made by make_golden.py from a
strict-IEEE build of /root/reference/d2q9-bgk.c) -- runs everywhere; (b) against that
reference build directly on the shipped decks and random states -- only where
oracle/_ref/ exists (the build container, or shipped prebuilt to the GPU box)."""
import numpy as np
import pytest

from conftest import KATS, deck_paths, load_kat


def _kat_param(O, k):
    return O.OrcParam(int(k["nx"]), int(k["ny"]), 10, int(k["reynolds_dim"]),
                      float(k["density"]), float(k["accel"]), float(k["omega"]))


@pytest.mark.parametrize("name", KATS)
def test_oracle_matches_reference_known_answers_bitwise(O, oracle, name):
    k = load_kat(name)
    prm = _kat_param(O, k)
    ob = np.ascontiguousarray(k["obstacles"], dtype=np.int32)
    a = k["cells0"].copy()
    b = np.empty_like(a)
    av = []
    for tt in range(1, 11):
        av.append(oracle.timestep(prm, a, b, ob))
        a, b = b, a
        if tt in (1, 2, 10):
            assert np.array_equal(a.view(np.uint32), k[f"cells_after_{tt}"].view(np.uint32)), f"step {tt}"
    assert np.array_equal(np.array(av, np.float32).view(np.uint32), k["av_vels"].view(np.uint32))
    assert np.float32(oracle.reynolds(prm, a, ob)) == k["reynolds_after_10"]


def test_known_answers_exercise_every_branch():
    """The vectors must contain blocked cells, open top/bottom rows (y-wrap live), and
    accelerate-row cells that the guard (f3-w1 > 0 ...) refuses."""
    for name in KATS:
        k = load_kat(name)
        ob, ny = k["obstacles"], int(k["ny"])
        assert ob.sum() > 0 and ob[0].sum() < ob.shape[1] and ob[ny - 1].sum() < ob.shape[1]
        a1 = np.float32(k["density"]) * np.float32(k["accel"]) / np.float32(9)
        row = k["cells0"][ny - 2]
        refused = (~ob[ny - 2].astype(bool)) & (row[:, 3] - a1 <= 0)
        assert refused.any()


needs_ref = pytest.mark.skipif(
    not __import__("lbm_oracle").ReferenceStrict.available(),
    reason="oracle/_ref/libd2q9_ref_strict.so not present (reference checkout absent)")


@needs_ref
@pytest.mark.parametrize("deck,nsteps", [("128x128", 300), ("128x256", 300), ("256x256", 100), ("1024x1024", 8)])
def test_oracle_equals_reference_on_shipped_decks_bitwise(O, oracle, deck, nsteps):
    ref = O.ReferenceStrict()
    pf, of = deck_paths(deck)
    prm = O.read_params(pf)
    rp = O.to_ref_param(prm)
    ob = O.read_obstacles(of, prm.nx, prm.ny)
    a = oracle.init_cells(prm, np.float32)
    ra = a.copy()
    b, rb = np.empty_like(a), np.empty_like(a)
    for _ in range(nsteps):
        av = oracle.timestep(prm, a, b, ob)
        rav = ref.timestep_new2(rp, ra, rb, ob)
        assert np.float32(av) == np.float32(rav)
        a, b, ra, rb = b, a, rb, ra
    assert np.array_equal(a.view(np.uint32), ra.view(np.uint32))
    assert np.float32(oracle.av_velocity(prm, a, ob)) == np.float32(ref.av_velocity(rp, ra, ob))
    assert np.float32(oracle.reynolds(prm, a, ob)) == np.float32(ref.calc_reynolds(rp, ra, ob))


@needs_ref
@pytest.mark.parametrize("nx,ny,seed", [(2, 2, 1), (3, 5, 2), (17, 9, 3), (40, 31, 4), (128, 7, 5)])
def test_oracle_equals_reference_on_random_lattices_bitwise(O, oracle, nx, ny, seed):
    """Ragged and minimum sizes (the peeled reference needs nx, ny >= 2), dense random obstacles."""
    ref = O.ReferenceStrict()
    rng = np.random.default_rng(seed)
    prm = O.OrcParam(nx, ny, 5, 7, 0.13, 0.02, 1.7)
    rp = O.to_ref_param(prm)
    ob = (rng.random((ny, nx)) < 0.3).astype(np.int32)
    ob[0, 0] = 0
    a = (0.05 + 0.1 * rng.random((ny, nx, 9))).astype(np.float32)
    ra = a.copy()
    b, rb = np.empty_like(a), np.empty_like(a)
    for _ in range(5):
        av = oracle.timestep(prm, a, b, ob)
        rav = ref.timestep_new2(rp, ra, rb, ob)
        assert np.float32(av) == np.float32(rav)
        a, b, ra, rb = b, a, rb, ra
    assert np.array_equal(a.view(np.uint32), ra.view(np.uint32))
