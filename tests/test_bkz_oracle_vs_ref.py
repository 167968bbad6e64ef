"""Pins oracle/gso_oracle.c::oracle_gso_bkz against the REAL reference:
BKZReduction<Z_NR<long>,FP_NR<double>>::bkz() (fplll/bkz.cpp:522-668, svp_reduction :274-358,
svp_postprocessing :126-272, tour/hkz :360-441) with empty strategies — BASELINE config 2's setting —
on tests/golden/bkz_*.json (oracle/ref_driver.cpp `bkzfix`).  The output basis, the status and the
total number of enumeration nodes must be identical."""
import os

import numpy as np
import pytest

import conftest as C


@pytest.mark.parametrize("path", C.bkz_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_bkz_oracle_matches_reference(path):
    f = C.load_bkz_fixture(path)
    g = C.OracleGSO(f["b_in"])
    st, info = g.bkz(f["block_size"], f["delta"], f["eta"], f["max_loops"], f["auto_abort"])
    assert st == f["status"]
    nodes = (int(info[1]) & 0xffffffff) | (int(info[2]) << 32)
    assert nodes == f["nodes"]
    assert np.array_equal(g.b, f["b_out"])
    assert not np.array_equal(f["b_in"], f["b_out"])
    g.close()


def test_config2_at_full_size():
    """BASELINE config 2 at its full size: BKZ-20 (no strategies, BKZ_DEFAULT, to convergence) on a
    120-dimensional q-ary lattice (gen_qary_prime(60, 20), LLL-reduced first): 138 tours, 1.03e7
    enumeration nodes, 4.3 s in the reference — basis, status and node count identical (~4 s)."""
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "c2_bkz20_q120.json.gz"))
    assert (f["d"], f["block_size"], f["max_loops"]) == (120, 20, 0)
    g = C.OracleGSO(f["b_in"])
    st, info = g.bkz(f["block_size"], f["delta"], f["eta"], f["max_loops"], f["auto_abort"])
    assert st == f["status"] == 1
    assert info[0] == 138
    assert ((int(info[1]) & 0xffffffff) | (int(info[2]) << 32)) == f["nodes"] == 10252068
    assert np.array_equal(g.b, f["b_out"])
    g.close()
