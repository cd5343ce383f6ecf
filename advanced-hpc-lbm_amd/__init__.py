"""advanced-hpc-lbm on MI355X: Python face of the C ABI (include/lbm_mi355x.h).

This package is plumbing for tests and bench.py: it binds
`liblbm_mi355x.so` (hand-written HIP kernels for gfx950 + the C ABI) with
ctypes and mirrors the names of the reference's own interface for the
time-step path (/root/reference/d2q9-bgk.c):

    t_param                      -> Param            (d2q9-bgk.c:64-73)
    initialise(param, obst)      -> initialise()     (d2q9-bgk.c:2716-2869)
    timestep_new2(...) + swap    -> timestep_new2()  (d2q9-bgk.c:98,182,190,228)
    main's step loop             -> Lattice.run()    (d2q9-bgk.c:180-201)
    av_velocity / calc_reynolds  -> Lattice.av_velocity() / .reynolds()
    write_values                 -> write_values()   (d2q9-bgk.c:2918-2999)

There is NO CPU fallback here: if the shared library is missing or no HIP
device is visible, loading / creating a lattice raises LbmError.  The CPU
oracle lives under oracle/ and is imported by tests only.

The directory name carries a hyphen, so `import advanced_hpc_lbm_amd` (the
one-file shim at the repo root) is the import path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LBM_MI355X_LIB") or os.path.join(_HERE, "liblbm_mi355x.so")   # (the override is for A/B timing of two builds)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "lbm_mi355x.h")

NSPEEDS = 9
EXCHANGE_AUTO, EXCHANGE_COPY, EXCHANGE_RCCL, EXCHANGE_P2P = 0, 1, 2, 3

# every symbol include/lbm_mi355x.h declares
ABI_SYMBOLS = (
    "lbm_last_error", "lbm_device_count", "lbm_create", "lbm_rccl_unique_id", "lbm_create_rank",
    "lbm_create_rank_ex", "lbm_p2p_handle", "lbm_p2p_connect",
    "lbm_slab_rows", "lbm_num_slabs", "lbm_run", "lbm_last_run_ms", "lbm_read_state",
    "lbm_av_velocity", "lbm_reynolds", "lbm_total_density", "lbm_final_state", "lbm_destroy",
    "lbm_timestep", "lbm_set_option", "lbm_get_info", "lbm_plan_tiles",
)


class LbmError(RuntimeError):
    pass


class Param(C.Structure):
    """Field-for-field the reference's t_param (d2q9-bgk.c:64-73) = lbm_param."""
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("maxIters", C.c_int),
                ("reynolds_dim", C.c_int), ("density", C.c_float),
                ("accel", C.c_float), ("omega", C.c_float)]

    def __repr__(self):
        return ("Param(nx=%d, ny=%d, maxIters=%d, reynolds_dim=%d, density=%g, accel=%g, omega=%g)"
                % (self.nx, self.ny, self.maxIters, self.reynolds_dim, self.density, self.accel, self.omega))


_lib = None


def load_library():
    """Loads liblbm_mi355x.so; raises LbmError if it has not been built.

    Import torch BEFORE calling this in a process that also uses torch: both
    link libamdhip64.so.7, and the first one loaded is the one shared.
    """
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LbmError(f"{LIB_PATH} is missing: run `make lib` (or __graft_entry__.build()); "
                       "there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, ip, fp, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)
    lib.lbm_last_error.restype = C.c_char_p
    lib.lbm_last_error.argtypes = []
    lib.lbm_device_count.argtypes = [ip]
    lib.lbm_create.argtypes = [C.POINTER(Param), vp, vp, C.c_int, vp, C.c_int, C.POINTER(vp)]
    lib.lbm_rccl_unique_id.argtypes = [vp]
    lib.lbm_create_rank.argtypes = [C.POINTER(Param), vp, vp, C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
    lib.lbm_create_rank_ex.argtypes = [C.POINTER(Param), vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(vp)]
    lib.lbm_p2p_handle.argtypes = [vp, vp]
    lib.lbm_p2p_connect.argtypes = [vp, vp, C.c_int]
    lib.lbm_slab_rows.argtypes = [vp, C.c_int, ip, ip]
    lib.lbm_num_slabs.argtypes = [vp]
    lib.lbm_run.argtypes = [vp, C.c_int, vp]
    lib.lbm_last_run_ms.argtypes = [vp, dp, dp]
    lib.lbm_read_state.argtypes = [vp, vp]
    lib.lbm_av_velocity.argtypes = [vp, fp]
    lib.lbm_reynolds.argtypes = [vp, fp]
    lib.lbm_total_density.argtypes = [vp, dp]
    lib.lbm_final_state.argtypes = [vp, vp]
    lib.lbm_destroy.argtypes = [vp]
    lib.lbm_timestep.argtypes = [C.POINTER(Param), vp, vp, vp, fp]
    lib.lbm_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    lib.lbm_get_info.argtypes = [vp, C.c_char_p, dp]
    for name in ABI_SYMBOLS:
        if name != "lbm_last_error":
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def _check(rc: int):
    if rc != 0:
        raise LbmError(f"[lbm error {rc}] {load_library().lbm_last_error().decode()}")


def plan_tiles(nx: int, rows: int, slabs_per_device: int = 1, compute_units: int = 256):
    """(rows per tile, rows per wavefront) of lbm_regtile's default tiling, or None where the lattice does not tile."""
    ty, r = C.c_int(0), C.c_int(0)
    rc = load_library().lbm_plan_tiles(nx, rows, slabs_per_device, compute_units, C.byref(ty), C.byref(r))
    return (ty.value, r.value) if rc == 0 else None


def device_count() -> int:
    n = C.c_int(0)
    _check(load_library().lbm_device_count(C.byref(n)))
    return n.value


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _check(load_library().lbm_rccl_unique_id(buf))
    return buf.raw


def _as_obstacles(obstacles, p: Param) -> np.ndarray:
    ob = np.ascontiguousarray(obstacles, dtype=np.int32)
    if ob.size != p.nx * p.ny:
        raise LbmError(f"obstacles has {ob.size} entries, lattice has {p.nx * p.ny} cells")
    return ob


def _as_cells(cells, p: Param):
    if cells is None:
        return None
    a = np.ascontiguousarray(cells, dtype=np.float32)
    if a.size != p.nx * p.ny * NSPEEDS:
        raise LbmError(f"cells has {a.size} floats, expected {p.nx * p.ny * NSPEEDS}")
    return a


class Lattice:
    """A lattice resident on the GPU(s): handle on an lbm_ctx."""

    def __init__(self, params: Param, obstacles, cells=None, nslabs: int = 1, devices=None,
                 exchange: int = EXCHANGE_AUTO, *, rank=None, nranks=None, device=0, unique_id=None):
        self._lib = load_library()
        self.params = params
        self._ctx = C.c_void_p()
        ob = _as_obstacles(obstacles, params)
        ce = _as_cells(cells, params)
        cp = ce.ctypes.data if ce is not None else None
        if rank is None:
            dv = None
            if devices is not None:
                dv = np.ascontiguousarray(devices, dtype=np.int32)
                if dv.size != nslabs:
                    raise LbmError("devices must list one HIP device per slab")
            _check(self._lib.lbm_create(C.byref(params), ob.ctypes.data, cp, nslabs,
                                        dv.ctypes.data if dv is not None else None, exchange,
                                        C.byref(self._ctx)))
        else:
            idbuf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
            if exchange == EXCHANGE_AUTO:
                _check(self._lib.lbm_create_rank(C.byref(params), ob.ctypes.data, cp, rank, nranks, device,
                                                 idbuf, C.byref(self._ctx)))
            else:
                _check(self._lib.lbm_create_rank_ex(C.byref(params), ob.ctypes.data, cp, rank, nranks, device,
                                                    idbuf, exchange, C.byref(self._ctx)))
        self.rank_mode = rank is not None

    # -- peer-to-peer halos without RCCL: the caller trades the handles ------------
    def p2p_handle(self) -> bytes:
        buf = C.create_string_buffer(64)
        _check(self._lib.lbm_p2p_handle(self._ctx, buf))
        return buf.raw

    def p2p_connect(self, handles):
        """handles: the 64-byte handles of all ranks, in rank order."""
        blob = b"".join(handles)
        buf = C.create_string_buffer(blob, len(blob))
        _check(self._lib.lbm_p2p_connect(self._ctx, buf, len(handles)))

    # -- the step loop -----------------------------------------------------
    def run(self, nsteps: int) -> np.ndarray:
        """nsteps x (timestep_new2 + swap); returns av_vels[nsteps] (float32)."""
        av = np.empty(max(nsteps, 0), dtype=np.float32)
        _check(self._lib.lbm_run(self._ctx, nsteps, av.ctypes.data))
        return av

    def last_run_ms(self):
        g, w = C.c_double(0), C.c_double(0)
        _check(self._lib.lbm_last_run_ms(self._ctx, C.byref(g), C.byref(w)))
        return g.value, w.value

    # -- state --------------------------------------------------------------
    def slab_rows(self, slab: int = 0):
        a, b = C.c_int(0), C.c_int(0)
        _check(self._lib.lbm_slab_rows(self._ctx, slab, C.byref(a), C.byref(b)))
        return a.value, b.value

    @property
    def num_slabs(self) -> int:
        return self._lib.lbm_num_slabs(self._ctx)

    def _local_rows(self) -> int:
        if self.rank_mode:
            a, b = self.slab_rows(0)
            return b - a
        return self.params.ny

    def read_state(self) -> np.ndarray:
        """Current lattice, reference AoS layout: (rows, nx, 9) float32."""
        out = np.empty((self._local_rows(), self.params.nx, NSPEEDS), dtype=np.float32)
        _check(self._lib.lbm_read_state(self._ctx, out.ctypes.data))
        return out

    def final_state(self) -> np.ndarray:
        """(rows, nx, 4) = u_x, u_y, |u|, pressure, computed on the GPU."""
        out = np.empty((self._local_rows(), self.params.nx, 4), dtype=np.float32)
        _check(self._lib.lbm_final_state(self._ctx, out.ctypes.data))
        return out

    def av_velocity(self) -> float:
        v = C.c_float(0)
        _check(self._lib.lbm_av_velocity(self._ctx, C.byref(v)))
        return v.value

    def reynolds(self) -> float:
        v = C.c_float(0)
        _check(self._lib.lbm_reynolds(self._ctx, C.byref(v)))
        return v.value

    def total_density(self) -> float:
        v = C.c_double(0)
        _check(self._lib.lbm_total_density(self._ctx, C.byref(v)))
        return v.value

    def set_option(self, key: str, value: int):
        _check(self._lib.lbm_set_option(self._ctx, key.encode(), value))

    def info(self, key: str) -> float:
        v = C.c_double(0)
        _check(self._lib.lbm_get_info(self._ctx, key.encode(), C.byref(v)))
        return v.value

    def close(self):
        if self._ctx:
            self._lib.lbm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def timestep_new2(params: Param, cells: np.ndarray, tmp_cells: np.ndarray, obstacles) -> float:
    """The reference's call shape (d2q9-bgk.c:98): one step on host arrays.

    Mutates `cells` (accelerate phase, row ny-2) and overwrites `tmp_cells`,
    exactly like the reference; the caller swaps.  Returns the average speed.
    """
    lib = load_library()
    if cells.dtype != np.float32 or tmp_cells.dtype != np.float32 or not cells.flags.c_contiguous \
            or not tmp_cells.flags.c_contiguous:
        raise LbmError("cells/tmp_cells must be C-contiguous float32 arrays")
    ob = _as_obstacles(obstacles, params)
    av = C.c_float(0)
    _check(lib.lbm_timestep(C.byref(params), cells.ctypes.data, tmp_cells.ctypes.data, ob.ctypes.data, C.byref(av)))
    return av.value


# ------------------------------------------------------------------ host I/O
def _die(message: str):
    raise LbmError(message)


def read_params(paramfile: str) -> Param:
    """Seven values `nx ny maxIters reynolds_dim density accel omega` (d2q9-bgk.c:2736-2762)."""
    try:
        tokens = open(paramfile).read().split()
    except OSError:
        _die(f"could not open input parameter file: {paramfile}")
    names = ("nx", "ny", "maxIters", "reynolds_dim", "density", "accel", "omega")
    vals = []
    for i, name in enumerate(names):
        try:
            vals.append(int(tokens[i]) if i < 4 else float(tokens[i]))
        except (IndexError, ValueError):
            _die(f"could not read param file: {name}")
    return Param(*vals)


def read_obstacles(obstaclefile: str, params: Param) -> np.ndarray:
    """Lines `x y 1` -> int32 (ny, nx) map, the reference's checks and messages (d2q9-bgk.c:2844-2857)."""
    try:
        text = open(obstaclefile).read()
    except OSError:
        _die(f"could not open input obstacles file: {obstaclefile}")
    ob = np.zeros((params.ny, params.nx), dtype=np.int32)
    tok = text.split()
    if len(tok) % 3:
        _die("expected 3 values per line in obstacle file")
    try:
        arr = np.array(tok, dtype=np.int64).reshape(-1, 3)
    except ValueError:
        _die("expected 3 values per line in obstacle file")
    if arr.size:
        if (arr[:, 0] < 0).any() or (arr[:, 0] > params.nx - 1).any():
            _die("obstacle x-coord out of range")
        if (arr[:, 1] < 0).any() or (arr[:, 1] > params.ny - 1).any():
            _die("obstacle y-coord out of range")
        if (arr[:, 2] != 1).any():
            _die("obstacle blocked value should be 1")
        ob[arr[:, 1], arr[:, 0]] = 1
    return ob


def initialise(paramfile: str, obstaclefile: str):
    """(params, cells, obstacles): parse both inputs, rest-equilibrium lattice (d2q9-bgk.c:2716-2869)."""
    p = read_params(paramfile)
    ob = read_obstacles(obstaclefile, p)
    d = np.float32(p.density)
    w = np.array([d * np.float32(4) / np.float32(9)] + [d / np.float32(9)] * 4 + [d / np.float32(36)] * 4,
                 dtype=np.float32)
    cells = np.broadcast_to(w, (p.ny, p.nx, NSPEEDS)).copy()
    return p, cells, ob


def write_values(params: Param, state4: np.ndarray, obstacles: np.ndarray, av_vels,
                 final_state_file="final_state.dat", av_vels_file="av_vels.dat"):
    """Both output files in the reference's line formats (d2q9-bgk.c:2978, 2993)."""
    ob = np.asarray(obstacles).reshape(params.ny, params.nx)
    s4 = np.asarray(state4, dtype=np.float32).reshape(params.ny, params.nx, 4)
    with open(final_state_file, "w") as fp:
        for jj in range(params.ny):
            fp.write("".join("%d %d %.12E %.12E %.12E %.12E %d\n" % (
                ii, jj, s4[jj, ii, 0], s4[jj, ii, 1], s4[jj, ii, 2], s4[jj, ii, 3], ob[jj, ii])
                for ii in range(params.nx)))
    with open(av_vels_file, "w") as fp:
        fp.write("".join("%d:\t%.12E\n" % (i, float(v)) for i, v in enumerate(av_vels)))


# ------------------------------------------------------------------ row slabs
# What crosses a slab boundary each step (the pull stencil, d2q9-bgk.c:971-998): a slab's
# bottom row is pulled into by the slab to the south through directions 4,7,8, its top row by
# the slab to the north through 2,5,6 -- three planes of one row per direction, 3*nx floats.
HALO_PLANES_TO_SOUTH = (4, 7, 8)   # of the sender's row 0
HALO_PLANES_TO_NORTH = (2, 5, 6)   # of the sender's last row
# Two steps per pass (lbm_sweep2) trade halos once per PAIR of steps: the adjacent row's cells are
# recomputed as the tile ring, which takes its centre planes 0,1,3, the three planes that stream
# across the boundary, and those three planes of the row behind it -- nine rows of nx floats
# (lbm_kernels.hip.h, kHaloSlots): (row offset from the sender's edge row, plane) per slot.
HALO9_TO_SOUTH = ((0, 0), (0, 1), (0, 3), (0, 4), (0, 7), (0, 8), (1, 4), (1, 7), (1, 8))
HALO9_TO_NORTH = ((0, 0), (0, 1), (0, 3), (0, 2), (0, 5), (0, 6), (1, 2), (1, 5), (1, 6))
TWO_STEP_TILE = (64, 16)           # lattice must tile: nx % 64 == 0, rows per slab % 16 == 0


def slab_bounds(ny: int, nslabs: int, slab: int):
    """Rows [begin, end) of slab `slab`: the library's own partition rule."""
    return slab * ny // nslabs, (slab + 1) * ny // nslabs


def ring_neighbours(rank: int, nranks: int):
    """(south, north) ranks of the periodic ring (the lattice wraps in y, d2q9-bgk.c:2132,2134)."""
    return (rank - 1) % nranks, (rank + 1) % nranks
