"""GPU parity tests for the batched Householder R-factor (fphip_hh_* through the C ABI): bit-exact
against golden vectors of the real reference and against the C oracle on the C3 / C5-sized inputs;
plus the reference's own cross-check (tests/test_gso.cpp:82-152): mu = R_ij/R_jj and r = R_ij*R_jj
agree with the GSO kernel's values."""
import os

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", C.hh_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_reference_fixture_parity(ctx, path):
    from fplll_amd.householder import MatHouseholderBatch
    f = C.load_hh_fixture(path)
    h = MatHouseholderBatch(ctx, 3, f["d"], f["n"], row_expo=bool(f["row_expo_on"]))
    h.set_basis(np.stack([f["b_in"]] * 3))
    assert list(h.update_R()) == [1, 1, 1]
    for L in range(3):
        R, e = h.get_R(L)
        assert np.array_equal(e, f["row_expo"])
        assert np.array_equal(np.tril(R[:, :f["d"]]), f["R"])
    h.close()


@pytest.mark.parametrize("d,row_expo", [(100, False), (180, True), (180, False)])
def test_wide_tilings_match_oracle(ctx, d, row_expo):
    from fplll_amd.householder import MatHouseholderBatch
    from fplll_amd.gso import _unreduced_copy
    import test_gso_gpu as T
    full = T._load_c3_basis()
    b = _unreduced_copy(full[:d, :], 2, 9)
    Ro, Vo, so, eo = C.oracle_hh_update_all(b, row_expo)
    h = MatHouseholderBatch(ctx, 64, d, full.shape[1], row_expo=row_expo)
    h.set_basis(b)
    h.broadcast_basis(0)
    assert int(h.update_R().min()) == 1
    for L in (0, 31, 63):
        R, e = h.get_R(L)
        assert np.array_equal(e, eo)
        assert np.array_equal(np.tril(R[:, :d]), np.tril(Ro[:, :d]))
    h.close()


def test_householder_agrees_with_gso(ctx):
    """test_gso.cpp:82-152: mu_ij = R_ij/R_jj, r_ij = R_ij*R_jj (1e-3 there; far tighter here)."""
    from fplll_amd.householder import MatHouseholderBatch
    from fplll_amd.gso import MatGSOBatch
    f = C.load_gso_fixture(os.path.join(C.GOLDEN, "gso_q48_p3.json"))
    b = f["b_out"]
    d = f["d"]
    h = MatHouseholderBatch(ctx, 1, d, d, row_expo=False)
    h.set_basis(b)
    h.update_R()
    R, _ = h.get_R(0)
    g = MatGSOBatch(ctx, 1, d, d, row_expo=False)
    g.set_basis(b)
    assert list(g.update_gso()) == [1]
    mu, r = g.get_mu_matrix(0), g.get_r_matrix(0)
    assert np.all(np.diag(R) > 0)
    for i in range(d):
        for j in range(i):
            assert abs(R[i, j] / R[j, j] - mu[i, j]) <= 1e-9 * max(1.0, abs(mu[i, j]))
            assert abs(R[i, j] * R[j, j] - r[i, j]) <= 1e-9 * max(1.0, abs(r[i, j]))
    h.close()
    g.close()
