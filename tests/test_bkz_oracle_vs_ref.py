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
