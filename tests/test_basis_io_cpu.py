"""Basis files and the auto-abort helper on the product side (fplll_amd.gso): save_basis_txt writes what the
reference's reader accepts (checked by letting the REAL reference read the file: `ref_driver gsoutil` must
return the Gram-Schmidt data of the original), load_basis_txt reads fplll's files (square and d x (d+1)),
BKZAutoAbort follows bkz.cpp:800-809."""
import json
import os
import subprocess

import numpy as np
import pytest

import conftest as C

DRV = os.path.join(C.ROOT, "oracle", "_ref", "ref_driver")


def test_round_trip_square_and_knapsack_shapes(tmp_path):
    from fplll_amd import gso as G
    b = G.load_basis_txt(os.path.join(C.GOLDEN, "basis_q180_seed0_lll_bkz20.txt"))
    assert b.shape == (180, 180)
    p = str(tmp_path / "b.txt")
    G.save_basis_txt(p, b)
    assert open(p).read() == open(os.path.join(C.GOLDEN, "basis_q180_seed0_lll_bkz20.txt")).read()
    assert np.array_equal(G.load_basis_txt(p), b)
    k = np.arange(5 * 6, dtype=np.int64).reshape(5, 6) - 11  # the shape of `latticegen r 5 ...`
    G.save_basis_txt(p, k)
    assert np.array_equal(G.load_basis_txt(p), k)
    assert G.load_basis_txt(os.path.join(C.GOLDEN, "basis_q200_seed7_lll.txt.gz")).shape == (200, 200)


def test_reference_reads_what_save_basis_txt_writes(tmp_path):
    if not os.path.exists(DRV):
        pytest.skip("oracle/_ref is not built on this machine")
    from fplll_amd import gso as G
    src = os.path.join(C.GOLDEN, "basis_q40_bkz10.txt")
    p = str(tmp_path / "copy.txt")
    G.save_basis_txt(p, G.load_basis_txt(src))
    outs = [subprocess.run([DRV, "gsoutil", f], capture_output=True, text=True, timeout=60) for f in (src, p)]
    assert all(o.returncode == 0 for o in outs)
    assert json.loads(outs[0].stdout) == json.loads(outs[1].stdout)


def test_bkz_auto_abort_counts_tours_without_progress():
    from fplll_amd.gso import BKZAutoAbort

    class Slopes:
        def __init__(self, seq):
            self.seq, self.calls = list(seq), []

        def get_current_slope(self, lattice, start, stop):
            self.calls.append((lattice, start, stop))
            return self.seq.pop(0)

    # get_current_slope is negative for a reduced basis; test_abort works on its negation
    m = Slopes([-0.060, -0.055, -0.056, -0.0551, -0.0552, -0.0553, -0.0554, -0.0555, -0.05])
    aa = BKZAutoAbort(m, 120, 3, lattice=2)
    got = [aa.test_abort() for _ in range(8)]
    # first call always resets; 0.055 improves; then five calls in a row that do not beat the best (0.055)
    assert got == [False, False, False, False, False, False, True, True]
    assert m.calls[0] == (2, 3, 120)
    # an improvement resets the count; `scale` loosens what counts as one
    assert aa.test_abort() is False and aa.no_dec == 0
    m2 = Slopes([-0.06, -0.0599, -0.0598])
    a2 = BKZAutoAbort(m2, 50)
    assert [a2.test_abort(0.99, 2), a2.test_abort(0.99, 2), a2.test_abort(0.99, 2)] == [False, False, True]
