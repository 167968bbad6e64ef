"""GPU parity tests for the batched HLLL kernel (fphip_hh_hlll through the C ABI): the reduced basis
and the status must equal the real reference's (tests/golden/hlll_*.json) and the C oracle's on
seeded inputs; afterwards R / row_expo must be the Householder R factor of the reduced basis."""
import os

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu


def _qary(rng, d, k, q):
    b = np.zeros((d, d), dtype=np.int64)
    b[:k, :k] = np.eye(k, dtype=np.int64)
    b[:k, k:] = rng.integers(0, q, size=(k, d - k))
    b[k:, k:] = q * np.eye(d - k, dtype=np.int64)
    return b


@pytest.mark.parametrize("path", C.hlll_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_reference_fixture_parity(ctx, path):
    from fplll_amd.householder import MatHouseholderBatch
    f = C.load_hlll_fixture(path)
    h = MatHouseholderBatch(ctx, 3, f["d"], f["n"], row_expo=True)
    h.set_basis(np.stack([f["b_in"]] * 3))
    st, info = h.hlll(f["delta"], f["eta"], f["theta"], f["c"])
    assert list(st) == [f["status"]] * 3
    out = h.get_basis(0, 3)
    for L in range(3):
        assert np.array_equal(out[L], f["b_out"])
    ost, ob, oinfo = C.oracle_hlll(f["b_in"], f["delta"], f["eta"], f["theta"], f["c"])
    assert list(info[0]) == list(oinfo)
    # R factor left behind = update_R() of the reduced basis (rows are final when hlll() ends)
    R, e = h.get_R(0)
    Ro, Vo, so, eo = C.oracle_hh_update_all(f["b_out"], True)
    assert np.array_equal(e, eo)
    assert np.array_equal(np.tril(R[:, :f["d"]]), np.tril(Ro[:, :f["d"]]))
    h.close()


@pytest.mark.parametrize("d,n_extra", [(2, 0), (3, 1), (17, 0), (33, 2), (64, 0), (65, 0), (90, 0)])
def test_seeded_vs_oracle_heterogeneous_batch(ctx, d, n_extra):
    from fplll_amd.householder import MatHouseholderBatch
    rng = np.random.default_rng(2000 + d)
    B = 5
    n = d + n_extra
    bs = []
    for L in range(B):
        if n_extra == 0 and d >= 4:
            b = _qary(rng, d, d // 2, int(rng.integers(50, 5000)))
        else:
            b = np.zeros((d, n), dtype=np.int64)
            b[:, :d] = np.eye(d, dtype=np.int64)
            lo = d - 1 if n_extra == 0 else d
            b[:, lo:] += rng.integers(-10**6, 10**6, size=(d, n - lo))
        bs.append(b)
    h = MatHouseholderBatch(ctx, B, d, n, row_expo=True)
    h.set_basis(np.stack(bs))
    st, info = h.hlll()
    out = h.get_basis(0, B)
    for L in range(B):
        ost, ob, oinfo = C.oracle_hlll(bs[L])
        assert st[L] == ost == 1
        assert list(info[L]) == list(oinfo)
        assert np.array_equal(out[L], ob)
    h.close()


def test_hlll_large_batch_stress(ctx):
    from fplll_amd.householder import MatHouseholderBatch
    rng = np.random.default_rng(12)
    d, B = 40, 1536
    base = [_qary(rng, d, d // 2, 1009 + 2 * i) for i in range(8)]
    h = MatHouseholderBatch(ctx, B, d, d, row_expo=True)
    h.set_basis(np.stack([base[i % 8] for i in range(B)]))
    st, info = h.hlll()
    assert np.all(st == 1)
    out = h.get_basis(0, B)
    for i in range(8):
        ost, ob, oinfo = C.oracle_hlll(base[i])
        assert ost == 1
        for L in range(i, B, 8):
            assert info[L][0] == oinfo[0]
            assert np.array_equal(out[L], ob), (i, L)
    h.close()
