"""Pins oracle/gso_oracle.c::oracle_gso_lll against the REAL reference:
LLLReduction<Z_NR<long>,FP_NR<double>>::lll on MatGSO(GSO_ROW_EXPO) (fplll/lll.cpp:44-164,
gso.cpp:289-366) — tests/golden/lll_*.json from oracle/ref_driver.cpp `lllfix`: raw q-ary, knapsack
(intrel) and uniform bases, leading zero rows and linearly dependent rows (the "zeros" path), a
sub-range (kappa_start > 0) and kappa_min > 0.  The output basis, the swap count, the number of
zero rows and the status must be identical."""
import os

import numpy as np
import pytest

import conftest as C


@pytest.mark.parametrize("path", C.lll_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_lll_oracle_matches_reference(path):
    f = C.load_lll_fixture(path)
    g = C.OracleGSO(f["b_in"])
    st, info = g.lll(f["kmin"], f["kstart"], f["kend"], f["delta"], f["eta"], flags=f["flags"])
    assert st == f["status"]
    assert info[1] == f["n_swaps"]
    assert info[2] == f["zeros"]
    assert info[0] == f["final_kappa"]
    assert np.array_equal(g.b, f["b_out"])
    assert not np.array_equal(f["b_in"], f["b_out"])
    g.close()


def test_inverse_transform_is_the_inverse_transpose_of_u():
    """u_inv_t (enable_inverse_transform, gso.cpp:84-158) of the reference runs that track it (`lll_*_u_uinv` fixtures,
    `LLLFIX_UINV=1`) equals the exact inverse transpose of their u — which is how this repo provides it
    (fplll_amd.gso.inverse_transpose / MatGSOBatch.get_inverse_transform_t: derived on the host from the device's u)."""
    import glob
    import json
    from fplll_amd.gso import inverse_transpose
    paths = sorted(glob.glob(os.path.join(C.GOLDEN, "lll_*_u_uinv.json")))
    assert len(paths) >= 2
    for p in paths:
        with open(p) as fh:
            j = json.load(fh)
        d = j["d"]
        u = np.array(j["u_out"], dtype=np.int64).reshape(d, d)
        want = np.array(j["u_inv_t_out"], dtype=object).reshape(d, d)
        got = inverse_transpose(u)
        assert np.array_equal(got, want), os.path.basename(p)
        assert np.array_equal(np.array(u, dtype=object).dot(got.T), np.eye(d, dtype=object))
    with pytest.raises(ValueError):
        inverse_transpose(np.array([[2, 0], [0, 1]]))
    with pytest.raises(ValueError):
        inverse_transpose(np.array([[1, 1], [1, 1]]))
