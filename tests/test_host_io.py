"""Host-side logic: the reference's input formats and error messages
(initialise, d2q9-bgk.c:2727-2857), output line formats (write_values, 2978/2993),
our own checker (reference check/check.py semantics) and the CLI's
argument / error convention (main 159-167, die/usage 3001-3013)."""
import io
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, deck_paths

EXE = os.path.join(ROOT, "d2q9-bgk")


def test_read_params_shipped_decks(L):
    p = L.read_params(deck_paths("1024x1024")[0])
    assert (p.nx, p.ny, p.maxIters, p.reynolds_dim) == (1024, 1024, 20000, 10)
    assert (p.density, p.accel, p.omega) == (np.float32(0.1), np.float32(0.01), np.float32(1.85))
    p = L.read_params(deck_paths("128x256")[0])
    assert (p.nx, p.ny, p.maxIters) == (128, 256, 40000)


def test_read_params_errors(L, tmp_path):
    with pytest.raises(L.LbmError, match="could not open input parameter file"):
        L.read_params(str(tmp_path / "nope.params"))
    f = tmp_path / "short.params"
    f.write_text("128\n128\n100\n")
    with pytest.raises(L.LbmError, match="could not read param file: reynolds_dim"):
        L.read_params(str(f))
    f.write_text("128\n128\n100\n10\n0.1\nabc\n1.85\n")
    with pytest.raises(L.LbmError, match="could not read param file: accel"):
        L.read_params(str(f))


def test_read_obstacles_counts_and_duplicates(L):
    """Blocked-cell counts of the shipped decks; duplicate lines allowed."""
    for deck, blocked in (("128x128", 508), ("256x256", 1020), ("1024x1024", 5114)):
        pf, of = deck_paths(deck)
        p = L.read_params(pf)
        ob = L.read_obstacles(of, p)
        assert ob.shape == (p.ny, p.nx) and int(ob.sum()) == blocked
    pf, of = deck_paths("128x256")
    p = L.read_params(pf)
    ob = L.read_obstacles(of, p)
    assert ob[0, 1:-1].sum() == 0 and ob[255, 1:-1].sum() == 0   # open top/bottom rows: y-wrap is live


def test_read_obstacles_errors(L, tmp_path):
    p = L.Param(8, 6, 1, 1, 0.1, 0.005, 1.85)
    f = tmp_path / "ob.dat"
    for text, msg in (("1 2\n", "expected 3 values per line"), ("8 0 1\n", "x-coord out of range"),
                      ("-1 0 1\n", "x-coord out of range"), ("0 6 1\n", "y-coord out of range"),
                      ("0 0 2\n", "blocked value should be 1")):
        f.write_text(text)
        with pytest.raises(L.LbmError, match=msg):
            L.read_obstacles(str(f), p)
    f.write_text("")
    assert L.read_obstacles(str(f), p).sum() == 0
    with pytest.raises(L.LbmError, match="could not open input obstacles file"):
        L.read_obstacles(str(tmp_path / "nope.dat"), p)


def test_initialise_is_rest_equilibrium(L, O, oracle):
    pf, of = deck_paths("128x128")
    p, cells, ob = L.initialise(pf, of)
    ref = oracle.init_cells(O.read_params(pf), np.float32)
    assert np.array_equal(cells.view(np.uint32), ref.view(np.uint32))


def test_write_values_line_formats(L, tmp_path):
    """Byte-identical to golden lines when fed the golden numbers."""
    gold = np.loadtxt(os.path.join(GOLDEN, "128x128.final_state.dat"), max_rows=128 * 3)
    p = L.Param(128, 3, 4, 10, 0.1, 0.005, 1.85)
    s4 = gold[:, 2:6].astype(np.float64).reshape(3, 128, 4)
    ob = gold[:, 6].astype(np.int32).reshape(3, 128)
    av = np.loadtxt(os.path.join(GOLDEN, "128x128.av_vels.dat"), usecols=[1], max_rows=4)
    fs, avf = tmp_path / "fs.dat", tmp_path / "av.dat"
    # float32 state would round the golden doubles: check the format with exactly representable values
    s4 = s4.astype(np.float32)
    L.write_values(p, s4, ob, av.astype(np.float32), str(fs), str(avf))
    lines = fs.read_text().splitlines()
    assert len(lines) == 384
    assert lines[0] == "0 0 0.000000000000E+00 0.000000000000E+00 0.000000000000E+00 %.12E 1" % np.float32(gold[0, 5])
    back = np.loadtxt(str(fs))
    assert np.array_equal(back[:, :2], gold[:, :2]) and np.array_equal(back[:, 6], gold[:, 6])
    assert np.allclose(back[:, 2:6], gold[:, 2:6], rtol=1e-6, atol=1e-12)
    alines = avf.read_text().splitlines()
    assert alines[0].startswith("0:\t") and alines[3].startswith("3:\t") and "E-0" in alines[0]


def test_checker_semantics(tmp_path):
    import check_results as CR
    ref = np.array([1.0, 2.0, 4.0])
    dev = CR.worst_deviation(ref, np.array([1.0, 2.0, 4.0]))
    assert dev["percent"] == 0 and dev["total"] == 0
    dev = CR.worst_deviation(ref, np.array([1.0, 2.02, 4.0]))     # 100*(2-2.02)/2.02
    assert dev["index"] == 1 and abs(dev["percent"] - 100 * (2 - 2.02) / 2.02) < 1e-12
    assert CR.passes(dev, 1.0) and not CR.passes(dev, 0.5)
    assert not CR.passes(CR.worst_deviation(ref, np.array([1.0, 0.0, 4.0])), 1.0)       # inf
    assert not CR.passes(CR.worst_deviation(np.array([0.0]), np.array([0.0])), 1.0)     # nan


def test_checker_reports_reference_float_binary_like_check_py():
    """Fixture: outputs of the reference binary as shipped (float, -Ofast) on 128x128.  The
    reference's own check.py printed max av_vels deviation 0.058 % at step 39882 and max
    pressure deviation -0.069 % at (95,62) for it."""
    import check_results as CR
    path = os.path.join(GOLDEN, "ref_float_128x128.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    with np.load(path) as z:
        av, pr = z["av_vels"].astype(np.float64), z["pressure"].astype(np.float64)
    gold_av = np.loadtxt(os.path.join(GOLDEN, "128x128.av_vels.dat"), usecols=[1])
    gold_fs = np.loadtxt(os.path.join(GOLDEN, "128x128.final_state.dat"), usecols=[0, 1, 5])
    a = CR.worst_deviation(gold_av, av)
    f = CR.worst_deviation(gold_fs[:, 2], pr.ravel())
    assert CR.passes(a, 1.0) and CR.passes(f, 1.0)
    assert 0.03 < abs(a["percent"]) < 0.1 and a["index"] > 39000
    assert 0.04 < abs(f["percent"]) < 0.1


def test_checker_end_to_end_on_files(tmp_path):
    import check_results as CR
    ga = os.path.join(GOLDEN, "128x128.av_vels.dat")
    gf = os.path.join(GOLDEN, "128x128.final_state.dat")
    out = io.StringIO()
    ok, a, f = CR.compare(ga, gf, ga, gf, out=out)
    assert ok and "Both tests passed!" in out.getvalue() and a["total"] == 0 and f["total"] == 0
    # a step-count mismatch and a coordinate mismatch must both fail
    short = tmp_path / "short.dat"
    short.write_text("".join(open(ga).readlines()[:100]))
    assert not CR.compare(ga, gf, str(short), gf, out=io.StringIO())[0]
    lines = open(gf).readlines()
    lines[5] = "9 9 " + lines[5].split(" ", 2)[2]
    bad = tmp_path / "bad.dat"
    bad.write_text("".join(lines))
    assert not CR.compare(ga, gf, ga, str(bad), out=io.StringIO())[0]
    assert CR.main(["--ref-av-vels-file", ga, "--ref-final-state-file", gf,
                    "--av-vels-file", ga, "--final-state-file", gf]) == 0


def test_reference_check_py_accepts_what_ours_accepts(tmp_path):
    """Where the reference checkout is present, its unchanged check.py agrees with ours."""
    ref_check = "/root/reference/check/check.py"
    if not os.path.exists(ref_check):
        pytest.skip("reference checkout absent")
    ga = os.path.join(GOLDEN, "128x128.av_vels.dat")
    gf = os.path.join(GOLDEN, "128x128.final_state.dat")
    r = subprocess.run(["python", ref_check, "--ref-av-vels-file", ga, "--ref-final-state-file", gf,
                        "--av-vels-file", ga, "--final-state-file", gf], capture_output=True, text=True)
    assert r.returncode == 0 and "Both tests passed!" in r.stdout


@pytest.mark.skipif(not os.path.exists(EXE), reason="d2q9-bgk not built (make)")
def test_cli_usage_and_die_convention(tmp_path):
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr == f"Usage: {EXE} <paramfile> <obstaclefile>\n"
    r = subprocess.run([EXE, "a", "b", "c"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Usage: ")
    r = subprocess.run([EXE, str(tmp_path / "none.params"), "x"], capture_output=True, text=True)
    assert r.returncode == 1
    err = r.stderr.splitlines()
    assert err[0].startswith("Error at line ") and " of file " in err[0] and err[0].endswith(":")
    assert err[1] == f"could not open input parameter file: {tmp_path / 'none.params'}"
    pf = tmp_path / "p.params"
    pf.write_text("8\n8\n2\n")
    r = subprocess.run([EXE, str(pf), "x"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.splitlines()[1] == "could not read param file: reynolds_dim"
    pf.write_text("8\n8\n2\n10\n0.1\n0.005\n1.85\n")
    of = tmp_path / "o.dat"
    of.write_text("8 0 1\n")
    r = subprocess.run([EXE, str(pf), str(of)], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.splitlines()[1] == "obstacle x-coord out of range"
    of.write_text("0 0 1\n1 1\n")
    r = subprocess.run([EXE, str(pf), str(of)], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.splitlines()[1] == "expected 3 values per line in obstacle file"
    r = subprocess.run([EXE, str(pf), str(tmp_path / "none.dat")], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.splitlines()[1].startswith("could not open input obstacles file: ")


def test_deck_generator_reproduces_the_shipped_1024_deck(tmp_path):
    """tools/make_deck.py at 1024 x 1024 = the shipped deck: identical params file, identical set of blocked
    cells (the shipped obstacle file lists a few cells twice); bench.py's in-memory map is the same function."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_deck
    pf, of, nb = make_deck.write_deck(1024, 1024, 20000, outdir=str(tmp_path))
    assert open(pf).read() == open(os.path.join(ROOT, "input_1024x1024.params")).read()
    mine = np.loadtxt(of, dtype=int)
    ref = np.loadtxt(os.path.join(ROOT, "obstacles_1024x1024.dat"), dtype=int)
    assert set(map(tuple, mine)) == set(map(tuple, ref)) and nb == 5114 and np.all(mine[:, 2] == 1)
    assert make_deck.wall_x(8192) == 2730
    big = make_deck.obstacle_map(8192, 64)
    assert big[:, 2730].all() and big[0].all() and big[-1].all() and big[:, 0].all() and big[:, -1].all()
    assert int(big[1:-1, 1:-1].sum()) == 62       # the wall only
