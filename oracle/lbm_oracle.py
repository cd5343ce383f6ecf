"""ctypes front-end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (advanced-hpc-lbm_amd/) never does.

It wraps
  * oracle/liblbm_oracle.so       -- our strict-IEEE restatement (float + double),
  * oracle/liblbm_oracle_fast.so  -- same source, reference Makefile flags (timing),
  * oracle/_ref/libd2q9_ref_strict.so -- the reference's own d2q9-bgk.c compiled
    with strict flags (only where oracle/Makefile could build it / it was shipped
    prebuilt); used to pin the restatement bit for bit,
and holds small numpy helpers for the reference's text formats
(/root/reference/d2q9-bgk.c:2736-2762 params, 2844-2857 obstacles,
2978/2993 output lines).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class OrcParam(C.Structure):
    """Mirror of orc_param in lbm_oracle.c (doubles: cast per flavour in C)."""
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("maxIters", C.c_int),
                ("reynolds_dim", C.c_int), ("density", C.c_double),
                ("accel", C.c_double), ("omega", C.c_double)]


class RefParam(C.Structure):
    """Mirror of the reference's t_param (d2q9-bgk.c:64-73)."""
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("maxIters", C.c_int),
                ("reynolds_dim", C.c_int), ("density", C.c_float),
                ("accel", C.c_float), ("omega", C.c_float)]


def build(quiet: bool = True) -> None:
    """(Re)build the oracle libraries and, where /root/reference exists, oracle/_ref."""
    subprocess.run(["make", "-C", HERE, "all"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


_libs: dict = {}


def _load(name: str):
    if name in _libs:
        return _libs[name]
    path = os.path.join(HERE, name)
    if not os.path.exists(path) and not name.startswith("_ref"):
        build()
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    _libs[name] = lib
    return lib


_F = {np.float32: ("f32", C.c_float), np.float64: ("f64", C.c_double)}


class Oracle:
    """dtype-generic handle on liblbm_oracle{,_fast}.so."""

    def __init__(self, flavour: str = "strict"):
        name = "liblbm_oracle.so" if flavour == "strict" else "liblbm_oracle_fast.so"
        self.lib = _load(name)
        if self.lib is None:
            raise RuntimeError(f"oracle library {name} could not be built")
        self.lib.orc_build_flavour.restype = C.c_char_p
        assert self.lib.orc_build_flavour().decode() == flavour
        for suf, ct in (("f32", C.c_float), ("f64", C.c_double)):
            for fn in ("orc_sweep_", "orc_timestep_", "orc_av_velocity_", "orc_reynolds_",
                       "orc_total_density_"):
                getattr(self.lib, fn + suf).restype = ct
            for fn in ("orc_init_cells_", "orc_accelerate_", "orc_run_", "orc_final_state_",
                       "orc_sweep_rows_", "orc_accelerate_row_"):
                getattr(self.lib, fn + suf).restype = None

    @staticmethod
    def _suf(dtype):
        return _F[np.dtype(dtype).type][0]

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(C.c_void_p)

    def init_cells(self, prm: OrcParam, dtype=np.float32) -> np.ndarray:
        cells = np.empty((prm.ny, prm.nx, 9), dtype=dtype)
        getattr(self.lib, "orc_init_cells_" + self._suf(dtype))(C.byref(prm), self._p(cells))
        return cells

    def accelerate(self, prm, cells, obstacles):
        getattr(self.lib, "orc_accelerate_" + self._suf(cells.dtype))(
            C.byref(prm), self._p(cells), self._p(obstacles))

    def sweep(self, prm, cells, tmp, obstacles) -> float:
        return getattr(self.lib, "orc_sweep_" + self._suf(cells.dtype))(
            C.byref(prm), self._p(cells), self._p(tmp), self._p(obstacles))

    def sweep_rows(self, prm, cells, tmp, obstacles, row_begin: int, row_end: int):
        """Sweep rows [row_begin, row_end) only -> (speed_sum, fluid_cells)."""
        ct = _F[np.dtype(cells.dtype).type][1]
        tot, cnt = ct(0), C.c_int(0)
        getattr(self.lib, "orc_sweep_rows_" + self._suf(cells.dtype))(
            C.byref(prm), self._p(cells), self._p(tmp), self._p(obstacles),
            C.c_int(row_begin), C.c_int(row_end), C.byref(tot), C.byref(cnt))
        return tot.value, cnt.value

    def accelerate_row(self, prm, cells, obstacles, row: int):
        getattr(self.lib, "orc_accelerate_row_" + self._suf(cells.dtype))(
            C.byref(prm), self._p(cells), self._p(obstacles), C.c_int(row))

    def timestep(self, prm, cells, tmp, obstacles) -> float:
        return getattr(self.lib, "orc_timestep_" + self._suf(cells.dtype))(
            C.byref(prm), self._p(cells), self._p(tmp), self._p(obstacles))

    def run(self, prm, cells, obstacles, nsteps: int) -> np.ndarray:
        """Advance `cells` in place by nsteps; returns av_vels[nsteps]."""
        assert cells.flags.c_contiguous and obstacles.dtype == np.int32
        tmp = np.empty_like(cells)
        av = np.empty(nsteps, dtype=cells.dtype)
        getattr(self.lib, "orc_run_" + self._suf(cells.dtype))(
            C.byref(prm), self._p(cells), self._p(tmp), self._p(obstacles),
            C.c_int(nsteps), self._p(av))
        return av

    def av_velocity(self, prm, cells, obstacles) -> float:
        return getattr(self.lib, "orc_av_velocity_" + self._suf(cells.dtype))(
            C.byref(prm), self._p(cells), self._p(obstacles))

    def reynolds(self, prm, cells, obstacles) -> float:
        return getattr(self.lib, "orc_reynolds_" + self._suf(cells.dtype))(
            C.byref(prm), self._p(cells), self._p(obstacles))

    def total_density(self, prm, cells) -> float:
        return getattr(self.lib, "orc_total_density_" + self._suf(cells.dtype))(
            C.byref(prm), self._p(cells))

    def final_state(self, prm, cells, obstacles) -> np.ndarray:
        """(ny, nx, 4) = u_x, u_y, |u|, pressure."""
        out = np.empty((prm.ny, prm.nx, 4), dtype=cells.dtype)
        getattr(self.lib, "orc_final_state_" + self._suf(cells.dtype))(
            C.byref(prm), self._p(cells), self._p(obstacles), self._p(out))
        return out


class ReferenceStrict:
    """The reference's own timestep_new2 / av_velocity (strict-flag build), if present."""

    def __init__(self):
        self.lib = _load(os.path.join("_ref", "libd2q9_ref_strict.so"))
        if self.lib is None:
            raise FileNotFoundError("oracle/_ref/libd2q9_ref_strict.so not built")
        vp = C.c_void_p
        self.lib.timestep_new2.restype = C.c_float
        self.lib.timestep_new2.argtypes = [RefParam, vp, vp, vp]
        self.lib.av_velocity.restype = C.c_float
        self.lib.av_velocity.argtypes = [RefParam, vp, vp]
        self.lib.calc_reynolds.restype = C.c_float
        self.lib.calc_reynolds.argtypes = [RefParam, vp, vp]

    @staticmethod
    def available() -> bool:
        return os.path.exists(os.path.join(HERE, "_ref", "libd2q9_ref_strict.so"))

    def timestep_new2(self, rp: RefParam, cells, tmp, obstacles) -> float:
        assert cells.dtype == np.float32 and obstacles.dtype == np.int32
        return self.lib.timestep_new2(rp, cells.ctypes.data, tmp.ctypes.data, obstacles.ctypes.data)

    def av_velocity(self, rp, cells, obstacles) -> float:
        return self.lib.av_velocity(rp, cells.ctypes.data, obstacles.ctypes.data)

    def calc_reynolds(self, rp, cells, obstacles) -> float:
        return self.lib.calc_reynolds(rp, cells.ctypes.data, obstacles.ctypes.data)


# ---------------------------------------------------------------- text formats

def read_params(path: str) -> OrcParam:
    """7 whitespace-separated tokens: nx ny maxIters reynolds_dim density accel omega."""
    tok = open(path).read().split()
    if len(tok) < 7:
        raise ValueError(f"could not read param file: {path}")
    return OrcParam(int(tok[0]), int(tok[1]), int(tok[2]), int(tok[3]),
                    float(tok[4]), float(tok[5]), float(tok[6]))


def to_ref_param(prm: OrcParam) -> RefParam:
    return RefParam(prm.nx, prm.ny, prm.maxIters, prm.reynolds_dim,
                    prm.density, prm.accel, prm.omega)


def read_obstacles(path: str, nx: int, ny: int) -> np.ndarray:
    """Lines 'x y 1' -> int32 (ny, nx) 0/1 map, with the reference's range checks."""
    obst = np.zeros((ny, nx), dtype=np.int32)
    data = np.loadtxt(path, dtype=np.int64, ndmin=2)
    if data.size:
        if data.shape[1] != 3:
            raise ValueError("expected 3 values per line in obstacle file")
        x, y, b = data[:, 0], data[:, 1], data[:, 2]
        if (x < 0).any() or (x > nx - 1).any():
            raise ValueError("obstacle x-coord out of range")
        if (y < 0).any() or (y > ny - 1).any():
            raise ValueError("obstacle y-coord out of range")
        if (b != 1).any():
            raise ValueError("obstacle blocked value should be 1")
        obst[y, x] = 1
    return obst


def read_av_vels(path: str) -> np.ndarray:
    return np.loadtxt(path, usecols=[1])


def read_final_state(path: str) -> np.ndarray:
    """Columns x y u_x u_y u pressure flag -> float64 (n, 7)."""
    return np.loadtxt(path)


def format_av_vels(av) -> str:
    """'%d:\\t%.12E\\n' per step (d2q9-bgk.c:2993)."""
    return "".join("%d:\t%.12E\n" % (i, float(v)) for i, v in enumerate(av))


def format_final_state(fs: np.ndarray, obstacles: np.ndarray) -> str:
    """'%d %d %.12E %.12E %.12E %.12E %d\\n' per cell, jj outer / ii inner (d2q9-bgk.c:2978);
    flag column = obstacles[jj, ii] as in the shipped goldens."""
    ny, nx = obstacles.shape
    lines = []
    for jj in range(ny):
        row = fs[jj]
        ob = obstacles[jj]
        for ii in range(nx):
            lines.append("%d %d %.12E %.12E %.12E %.12E %d\n" % (
                ii, jj, row[ii, 0], row[ii, 1], row[ii, 2], row[ii, 3], ob[ii]))
    return "".join(lines)
