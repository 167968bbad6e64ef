"""GPU parity tests for the batched BKZ kernel (fphip_gso_bkz through the C ABI): the reduced
basis, the status and the total number of enumeration nodes (fplll rule) must equal the real
reference's BKZReduction::bkz() with empty strategies (tests/golden/bkz_*.json) and the C oracle's
on seeded inputs.  One launch runs every tour: LLL sweeps, block enumerations, insertions."""
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


def _nodes(info_row):
    return (int(info_row[1]) & 0xffffffff) | ((int(info_row[2]) & 0xffffffff) << 32)


@pytest.mark.parametrize("path", C.bkz_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_reference_fixture_parity(ctx, path):
    from fplll_amd.gso import MatGSOBatch
    f = C.load_bkz_fixture(path)
    g = MatGSOBatch(ctx, 3, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * 3))
    st, info = g.bkz(f["block_size"], f["delta"], f["eta"], f["max_loops"], f["auto_abort"])
    assert list(st) == [f["status"]] * 3
    out = g.get_basis(0, 3)
    for L in range(3):
        assert np.array_equal(out[L], f["b_out"])
        assert _nodes(info[L]) == f["nodes"]
    g.close()


@pytest.mark.parametrize("d,beta", [(20, 6), (33, 10), (48, 12), (66, 8)])
def test_seeded_vs_oracle_heterogeneous_batch(ctx, d, beta):
    """LLL then BKZ on DIFFERENT lattices (device LLL feeds device BKZ, oracle LLL feeds oracle BKZ)."""
    from fplll_amd.gso import MatGSOBatch
    rng = np.random.default_rng(3000 + d)
    B = 5
    bs = [_qary(rng, d, d // 2, int(rng.integers(200, 20000))) for _ in range(B)]
    g = MatGSOBatch(ctx, B, d, d)
    g.set_basis(np.stack(bs))
    st, _ = g.lll()
    assert np.all(st == 1)
    st, info = g.bkz(beta)
    out = g.get_basis(0, B)
    for L in range(B):
        o = C.OracleGSO(bs[L])
        ost, _ = o.lll()
        assert ost == 1
        o2 = C.OracleGSO(o.b)
        bst, binfo = o2.bkz(beta)
        assert st[L] == bst == 1
        assert info[L][0] == binfo[0]
        assert _nodes(info[L]) == ((int(binfo[1]) & 0xffffffff) | ((int(binfo[2]) & 0xffffffff) << 32))
        assert np.array_equal(out[L], o2.b)
        o.close()
        o2.close()
    g.close()


def test_max_loops_status(ctx):
    """BKZ_MAX_LOOPS: one tour of a basis that needs more returns RED_BKZ_LOOPS_LIMIT (8)."""
    from fplll_amd.gso import MatGSOBatch
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "bkz_q100_b20_autoabort.json"))
    g = MatGSOBatch(ctx, 2, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * 2))
    st, info = g.bkz(f["block_size"], max_loops=1)
    o = C.OracleGSO(f["b_in"])
    ost, oinfo = o.bkz(f["block_size"], max_loops=1)
    assert list(st) == [ost, ost] == [8, 8]
    assert int(info[0][0]) == int(oinfo[0]) == 1
    assert np.array_equal(g.get_basis(0, 1)[0], o.b)
    o.close()
    g.close()


def test_bkz_improves_and_is_fixed_point(ctx):
    """size-independent properties: same lattice (determinant), first vector not longer than LLL's,
    and a BKZ-reduced basis is a fixed point (one clean tour, no change)."""
    from fplll_amd.gso import MatGSOBatch
    rng = np.random.default_rng(5)
    d, beta = 50, 14
    bs = np.stack([_qary(rng, d, d // 2, 4001 + 2 * i) for i in range(4)])
    g = MatGSOBatch(ctx, 4, d, d)
    g.set_basis(bs)
    st, _ = g.lll()
    assert np.all(st == 1)
    lll_out = g.get_basis(0, 4)
    st, info = g.bkz(beta)
    assert np.all(st == 1)
    out = g.get_basis(0, 4)
    for L in range(4):
        n_lll = float(np.sum(lll_out[L][0].astype(np.float64) ** 2))
        n_bkz = float(np.sum(out[L][0].astype(np.float64) ** 2))
        assert n_bkz <= n_lll
        r = g.get_r_matrix(L)
        e = g.row_expo(L).astype(np.float64)
        logdet = np.sum(np.log(np.diag(r)) + 2 * e * np.log(2.0)) / 2
        assert abs(logdet - (d - d // 2) * np.log(4001 + 2 * L)) < 1e-6
    st2, info2 = g.bkz(beta)
    assert np.all(st2 == 1)
    assert list(info2[:, 0]) == [1] * 4
    assert np.array_equal(g.get_basis(0, 4), out)
    g.close()
