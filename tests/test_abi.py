"""The C-ABI shared library loads and exports every symbol include/lbm_mi355x.h declares
(no compute without a GPU), and fails loudly -- never falls back -- when no device exists."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lbm_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lbm_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(L):
    assert _declared_symbols() == sorted(L.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(L):
    lib = C.CDLL(L.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_param_struct_matches_reference_t_param(L):
    """t_param = 4 ints + 3 floats, 28 bytes (d2q9-bgk.c:64-73)."""
    assert C.sizeof(L.Param) == 28
    assert [f[0] for f in L.Param._fields_] == ["nx", "ny", "maxIters", "reynolds_dim", "density", "accel", "omega"]


def test_no_device_is_an_error_not_a_fallback(L):
    if L.device_count() > 0:
        pytest.skip("a HIP device is visible here")
    p = L.Param(8, 8, 1, 1, 0.1, 0.005, 1.85)
    with pytest.raises(L.LbmError, match="no HIP device"):
        L.Lattice(p, np.zeros((8, 8), np.int32))
    cells = np.zeros((8, 8, 9), np.float32)
    with pytest.raises(L.LbmError, match="no HIP device"):
        L.timestep_new2(p, cells, cells.copy(), np.zeros((8, 8), np.int32))


def test_bad_arguments_are_rejected(L):
    lib = L.load_library()
    ctx = C.c_void_p()
    p = L.Param(8, 1, 1, 1, 0.1, 0.005, 1.85)   # ny = 1: accelerate row ny-2 does not exist
    ob = np.zeros(8, np.int32)
    assert lib.lbm_create(C.byref(p), ob.ctypes.data, None, 1, None, 0, C.byref(ctx)) == 1
    assert b"at least" in lib.lbm_last_error()
    p = L.Param(8, 8, 1, 1, 0.1, 0.005, 1.85)
    assert lib.lbm_create(C.byref(p), None, None, 1, None, 0, C.byref(ctx)) == 1
    assert lib.lbm_create(C.byref(p), ob.ctypes.data, None, 0, None, 0, C.byref(ctx)) == 1
    assert lib.lbm_run(None, 1, None) == 1


def test_product_never_touches_the_oracle():
    """Nothing under advanced-hpc-lbm_amd/ or the CLI host may reference oracle/."""
    pkg = os.path.join(ROOT, "advanced-hpc-lbm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp", ".inc")):
                text = open(os.path.join(dirpath, f)).read()
                assert "lbm_oracle" not in text and "liblbm_oracle" not in text, f
                assert not re.search(r"^\s*(import|from)\s+oracle", text, flags=re.M), f


def test_hand_issued_mail_loads_of_the_register_tile_kernel_are_left_alone_by_the_compiler():
    """lbm_regtile's asynchronous loop issues its mail loads as inline asm and retires them with counted waits; the
    destination registers belong to the compiler from the end of the asm statement although the data lands later
    (cdna_hip_programming.md 5.7).  tools/audit_regtile_isa.py compiles the library's device code and walks the ISA from
    every such load, along every path, to the wait that retires it: no instruction in between may touch the destination
    registers, no scratch, no AGPR traffic, no vector-written scalar operand within five instructions of an asm memory
    operation.  (Runs without a GPU: hipcc cross-compiles.)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_regtile_isa.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 finding(s)" in r.stdout and "asm loads audited" in r.stdout


def test_register_tiling_rule(L):
    """lbm_plan_tiles (host arithmetic, no GPU): the default tiling of lbm_regtile as measured on MI355X
    (profiles/r03_regtile_tilings.log) -- the shortest tiles that fit one per CU, but not one-row tiles; at most eight
    waves per tile where the lattice allows it, sixteen where it does not; slabs sharing a device share its CUs."""
    assert L.plan_tiles(1024, 1024) == (64, 4)          # the shipped deck: 256 tiles of 16 waves x 4 rows
    assert L.plan_tiles(1024, 512) == (32, 4)           # one rank's slab at N = 2 / 4 / 8
    assert L.plan_tiles(1024, 256) == (16, 2)
    assert L.plan_tiles(1024, 128) == (8, 1)
    assert L.plan_tiles(256, 256) == (4, 1) and L.plan_tiles(128, 128) == (2, 1) and L.plan_tiles(128, 256) == (2, 1)
    assert L.plan_tiles(64, 8) == (2, 1) and L.plan_tiles(64, 1) == (1, 1)
    assert L.plan_tiles(1024, 128, slabs_per_device=8) == (64, 4)     # eight slabs on ONE device: the undivided lattice's tiles
    assert L.plan_tiles(1024, 256, slabs_per_device=2) == (32, 4)
    assert L.plan_tiles(1000, 1000) is None and L.plan_tiles(48, 64) is None      # width no multiple of 64
    assert L.plan_tiles(2048, 2048) is None and L.plan_tiles(1024, 2048) is None  # more cells than the register file takes
    assert L.plan_tiles(1024, 1024, compute_units=128) is None
    assert L.plan_tiles(1024, 1000) is None             # 1000 rows: no tile height up to 64 divides them into <= 256 tiles ... (8 x 125)
