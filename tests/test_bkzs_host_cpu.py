"""CPU-side check of the PRODUCT's host logic for BKZ with strategies (no GPU involved): the two
decisions the calling thread serves to the waves through their mailboxes
(fplll_amd/csrc/gso_host.hip: serve_radius, serve_plan).

* radius + pruning set of a block (bkz.cpp:309-325, get_root_det gso_interface.cpp:220-242,
  adjust_radius_to_gh_bound :260-276, Strategy::get_pruning bkz_param.cpp:64-80) against the oracle
  (which is pinned to the reference on the same fixtures) — bit for bit, for every block of a sweep;
* the rerandomisation plan (rerandomize_block, bkz.cpp:43-80) against a direct restatement of the
  reference's loop drawing from the same GMP stream."""
import ctypes
import os

import numpy as np
import pytest

import conftest as C


class Strat(ctypes.Structure):
    _fields_ = [("max_block_size", ctypes.c_int), ("pre_off", ctypes.c_void_p),
                ("pre", ctypes.c_void_p), ("prune_off", ctypes.c_void_p),
                ("prune_gh", ctypes.c_void_p), ("prune_exp", ctypes.c_void_p),
                ("coeff_off", ctypes.c_void_p), ("coeff", ctypes.c_void_p)]


def strat_struct(S):
    keep = []

    def arr(key, dt):
        a = np.ascontiguousarray(S[key], dtype=dt)
        if a.size == 0:
            a = np.zeros(1, dtype=dt)
        keep.append(a)
        return a.ctypes.data
    st = Strat(int(S["max_block_size"]), arr("pre_off", np.int32), arr("pre", np.int32),
               arr("prune_off", np.int32), arr("prune_gh", np.float64), arr("prune_exp", np.float64),
               arr("coeff_off", np.int32), arr("coeff", np.float64))
    return st, keep


def product_lib():
    from fplll_amd import _lib
    return _lib.load()  # loads without a GPU; only device entry points need one


@pytest.mark.parametrize("name", ["bkzs_q64_b40_pre_gh", "bkzs_q64_b40_rerand", "bkzs_r40_b32_rerand"])
def test_radius_and_pruning_choice_match_oracle(name):
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, name + ".json"))
    S = f["strategies"]
    st, keep = strat_struct(S)
    lib, olib = product_lib(), C.oracle_lib()
    olib.oracle_gso_bkz_radius.restype = None
    olib.oracle_gso_bkz_radius.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                           ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    lib.fphip_debug_bkz_radius.restype = ctypes.c_int
    lib.fphip_debug_bkz_radius.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int),
                                           ctypes.POINTER(ctypes.c_double)]
    # two states of the lattice: the LLL-reduced input and the reference's BKZ output
    checked = 0
    seen = set()
    for basis in (f["b_in"], f["b_out"]):
        g = C.OracleGSO(basis)
        assert g.update_all()
        d = f["d"]
        rd = np.ascontiguousarray(np.diag(g.r))
        ex = np.asarray(g.row_expo, dtype=np.int64)
        for bs in range(2, f["block_size"] + 1, 3):
            for kappa in range(0, d - bs + 1, 5):
                # 0x20000: a dual block (self-dual BKZ): radius from 1 / r of the block's last row
                for flags, ghf in ((0x80, 1.1), (0x0, 1.1), (0x80, 1.3), (0x20080, 1.1), (0x20000, 1.1)):
                    omd, opr = ctypes.c_double(), ctypes.c_int()
                    olib.oracle_gso_bkz_radius(g.h, kappa, bs, flags, f["delta"], ghf, ctypes.byref(st),
                                               ctypes.byref(omd), ctypes.byref(opr))
                    r = np.ascontiguousarray(rd[kappa:kappa + bs])
                    e2 = np.ascontiguousarray(2 * ex[kappa:kappa + bs], dtype=np.int32)
                    pmd, ppr, pex = ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
                    rc = lib.fphip_debug_bkz_radius(ctypes.byref(st), ghf, bs, flags, f["delta"],
                                                    r.ctypes.data, e2.ctypes.data, ctypes.byref(pmd),
                                                    ctypes.byref(ppr), ctypes.byref(pex))
                    assert rc == 0
                    assert pmd.value.hex() == omd.value.hex(), (kappa, bs, flags)
                    assert ppr.value == opr.value
                    assert pex.value == S["prune_exp"][opr.value]
                    seen.add(opr.value)
                    checked += 1
        g.close()
    assert checked > 300
    assert len(seen) > 10  # many different pruning sets were chosen on the way


def reference_plan(rnd, lo, hi, density):
    """rerandomize_block(min_row, max_row, density), bkz.cpp:43-80, as a list of packed operations."""
    moves, ops = [], []
    if hi - lo <= 2:  # (== 2: the reference's `while (b == a)` cannot end; the product draws nothing)
        return moves, ops
    for _ in range(4 * (hi - lo)):
        a = rnd(0, hi - lo - 1) + lo
        b = a
        while b == a:
            b = rnd(0, hi - lo - 1) + lo
        moves.append(b | (a << 8))
    for a in range(lo, hi - 2):
        for _ in range(density):
            b = rnd(0, hi - (a + 1) - 1) + a + 1
            ops.append(a | (b << 8) | ((1 if rnd(0, 2) else 0) << 16))
    return moves, ops


@pytest.mark.parametrize("lo,hi", [(1, 40), (17, 19), (5, 68), (30, 33), (3, 4)])
def test_rerandomisation_plan_matches_reference_loop(lo, hi):
    lib = product_lib()
    (fn, user), draws = C.gmp_streams_native(2, 99)
    lib.fphip_debug_bkz_plan.restype = ctypes.c_int
    lib.fphip_debug_bkz_plan.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    py = C.GmpStreams(1, 99)
    for rep in range(3):  # consecutive calls continue the same stream
        plan = np.zeros(448, dtype=np.uint32)
        nm, no = ctypes.c_int(), ctypes.c_int()
        rc = lib.fphip_debug_bkz_plan(fn, user, 1, lo, hi, 3, plan.ctypes.data, ctypes.byref(nm),
                                      ctypes.byref(no))
        assert rc == 0
        moves, ops = reference_plan(py, lo, hi, 3)
        assert nm.value == len(moves) and no.value == len(ops)
        assert list(plan[:nm.value]) == moves
        assert list(plan[nm.value:nm.value + no.value]) == ops
    assert draws() == py.draws
