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


@pytest.mark.parametrize("src,d", [("q48", 48), ("c3", 100), ("c3", 180), ("q200", 200), ("c5", 256),
                                   ("q64raw", 64)])
def test_blocked_mfma_mode_agrees_with_exact_mode(ctx, src, d):
    """fphip_hh_update_R_blocked (compact-WY panels on v_mfma_f64_16x16x4_f64, hh_blocked.hip): the
    opt-in fast mode.  Row exponents and signs identical to the exact mode.  On REDUCED bases (what
    the R-factor is computed on inside HLLL / BKZ) R — hence mu = R_ij/R_jj and r = R_ij R_jj
    (tests/test_gso.cpp:82-152) — agrees within 1e-9 relative, the north-star tolerance for
    double-precision mu / r (observed ~1e-13).  On a raw, ill-conditioned q-ary basis the two modes
    are two equally valid roundings of the same factorisation: each row of R agrees to 1e-10 of the
    row's norm (backward error), which is all either of them promises there."""
    from fplll_amd.householder import MatHouseholderBatch
    from fplll_amd.gso import _unreduced_copy, load_basis_txt
    import test_gso_gpu as T
    reduced = True
    if src == "q48":
        b = C.load_gso_fixture(os.path.join(C.GOLDEN, "gso_q48_p3.json"))["b_out"]
    elif src == "c3":
        b = T._load_c3_basis()[:d, :]
    elif src == "q200":
        b = load_basis_txt(os.path.join(C.GOLDEN, "basis_q200_seed7_lll.txt.gz"))
    elif src == "c5":  # (beyond plain doubles for mu / r: babai fails on it in the reference too)
        b, reduced = C.load_hlll_fixture(os.path.join(C.GOLDEN, "c5_hlll_n256_double.json.gz"))["b_out"], False
    else:
        b, reduced = C.load_gso_fixture(os.path.join(C.GOLDEN, "gso_q64_p5.json"))["b_in"], False
    n = b.shape[1]
    for row_expo in (True, False):
        h = MatHouseholderBatch(ctx, 5, d, n, row_expo=row_expo)
        h.set_basis(np.stack([b] * 5))
        assert list(h.update_R()) == [1] * 5
        Re, ee = h.get_R(3)
        ms_exact = h.last_kernel_ms
        assert list(h.update_R(blocked=True)) == [1] * 5
        ms_blk = h.last_kernel_ms
        worst = 0.0
        for L in (0, 3, 4):
            Rb, eb = h.get_R(L)
            assert np.array_equal(eb, ee)
            Re_t, Rb_t = np.tril(Re[:, :d]), np.tril(Rb[:, :d])
            de, db = np.diag(Re_t), np.diag(Rb_t)
            assert np.all(db > 0)
            rown = np.sqrt((Re_t ** 2).sum(axis=1))
            assert np.all(np.abs(Rb_t - Re_t) <= 1e-10 * rown[:, None])
            if reduced:
                assert np.all(np.abs(db - de) <= 1e-9 * np.abs(de))
                mu_e, mu_b = Re_t / de[None, :], Rb_t / db[None, :]
                assert np.all(np.abs(mu_b - mu_e) <= 1e-9 * np.maximum(1.0, np.abs(mu_e)))
                r_e, r_b = Re_t * de[None, :], Rb_t * db[None, :]
                scale = np.maximum(np.abs(r_e), np.outer(de, de))
                assert np.all(np.abs(r_b - r_e) <= 1e-9 * scale)
                worst = max(worst, float(np.max(np.abs(mu_b - mu_e))))
        C.note(lambda: ("blocked vs exact R-factor %s %dx%d row_expo=%d: max |dmu| %.2e; kernel %.2f ms vs %.2f ms (batch 5)"
              % (src, d, n, row_expo, worst, ms_blk, ms_exact),))
        h.close()


@pytest.mark.parametrize("batch,d,n", [(1, 20, 20), (5, 48, 60), (17, 64, 64), (33, 65, 65), (16, 100, 128),
                                       (7, 129, 129), (40, 150, 180), (19, 192, 192)])
def test_rows_kernel_equals_the_column_kernel_bit_for_bit(ctx, batch, d, n, monkeypatch):
    """hh_rows.hip (a lane owns a ROW of one lattice and runs the reference's scalar loops; the kernel of rows up
    to 192 columns) against hh_update_kernel (lane = column, FPHIP_HH_ROWS=0) and the C oracle: the same bits in R —
    the reflectors' contribution above the diagonal included, where the two kernels keep what update_R leaves
    there — and the same row exponents, on ragged shapes (panels that end inside a tile, groups of 16 lattices that
    the batch does not fill, d != n) with different lattices in one launch, with and without row exponents."""
    from fplll_amd.householder import MatHouseholderBatch
    rng = np.random.default_rng(1000 * batch + d)
    bs = np.zeros((batch, d, n), dtype=np.int64)
    for L in range(batch):
        q = int(rng.integers(1 << 10, 1 << 30))
        k = d // 2
        bs[L, :k, :k] = np.eye(k, dtype=np.int64)
        bs[L, :k, k:] = rng.integers(0, q, size=(k, n - k))
        bs[L, k:, :] = rng.integers(-q, q, size=(d - k, n))
        bs[L, k:, :k] = 0
        for r in range(k, d):
            bs[L, r, min(n - 1, r)] += q
    for row_expo in (True, False):
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("FPHIP_HH_ROWS", mode)
            h = MatHouseholderBatch(ctx, batch, d, n, row_expo=row_expo)
            h.set_basis(bs)
            assert list(h.update_R()) == [1] * batch
            out[mode] = [h.get_R(L) for L in range(batch)]
            h.close()
        for L in range(batch):
            (R1, e1), (R0, e0) = out["1"][L], out["0"][L]
            assert np.array_equal(e1, e0), (L, row_expo)
            assert np.array_equal(R1.view(np.uint64), R0.view(np.uint64)), (L, row_expo)
        for L in (0, batch - 1):
            Ro, Vo, so, eo = C.oracle_hh_update_all(bs[L], row_expo)
            R1, e1 = out["1"][L]
            assert np.array_equal(e1, eo)
            assert np.array_equal(np.tril(R1[:, :d]).view(np.uint64), np.tril(Ro[:, :d]).view(np.uint64))


@pytest.mark.parametrize("path", C.hhsr_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_size_reduce_reference_fixture_parity(ctx, path, monkeypatch):
    """fphip_hh_size_reduce = MatHouseholder::size_reduce(kappa, end, start) (householder.cpp:402-451) on the state
    update_R() left, against the `hhsr` fixtures of the real reference: the returned flag, row kappa of the basis and
    the kappa + 1 leading entries of R[kappa] bit for bit, every other row untouched; from both R-factor kernels
    (the tails right of the diagonal that :551-556 fold into R[kappa] are the same in both)."""
    from fplll_amd.householder import MatHouseholderBatch
    f = C.load_hhsr_fixture(path)
    k = f["kappa"]
    for rows_kernel in ("1", "0"):
        monkeypatch.setenv("FPHIP_HH_ROWS", rows_kernel)
        h = MatHouseholderBatch(ctx, 5, f["d"], f["n"], row_expo=bool(f["row_expo_on"]))
        h.set_basis(np.stack([f["b_in"]] * 5))
        assert list(h.update_R()) == [1] * 5
        R0, _ = h.get_R(2)
        red, st = h.size_reduce(k, f["end"], f["start"])
        assert list(red) == [f["reduced"]] * 5 and list(st) == [1] * 5
        b = h.get_basis(0, 5)
        for L in range(5):
            assert np.array_equal(b[L, k], f["b_row"])
            others = np.arange(f["d"]) != k
            assert np.array_equal(b[L][others], f["b_in"][others])
            R, e = h.get_R(L)
            assert np.array_equal(R[k, :k + 1], f["R_row"][:k + 1])
            assert np.array_equal(R[others], R0[others]) and np.array_equal(e, f["row_expo"])
        h.close()


@pytest.mark.parametrize("d,row_expo,k,end,start", [(180, True, 170, 170, 0), (180, False, 100, 90, 17), (130, True, 129, 129, 64)])
def test_size_reduce_matches_oracle_at_size(ctx, d, row_expo, k, end, start):
    """The same on config 3's basis (180 columns: three registers per lane, rows of R and b streamed from HBM) against
    the C oracle, with distinct lattices in the batch."""
    from fplll_amd.householder import MatHouseholderBatch
    from fplll_amd.gso import _unreduced_copy
    import test_gso_gpu as T
    full = T._load_c3_basis()
    bs = [_unreduced_copy(full[:d, :], 2 + L, 9 + L) for L in range(4)]
    h = MatHouseholderBatch(ctx, 4, d, full.shape[1], row_expo=row_expo)
    h.set_basis(np.stack(bs))
    assert int(h.update_R().min()) == 1
    red, st = h.size_reduce(k, end, start)
    bo = h.get_basis(0, 4)
    for L in range(4):
        flag, b1, R1, e1 = C.oracle_hh_size_reduce(bs[L], row_expo, k, end, start)
        assert flag == int(red[L]) and int(st[L]) == 1
        assert np.array_equal(bo[L], b1)
        R, e = h.get_R(L)
        assert np.array_equal(R[k, :k + 1], R1[k, :k + 1]) and np.array_equal(e, e1)
    assert int(red.max()) == 1
    h.close()


def test_size_reduce_refuses_bad_ranges(ctx):
    from fplll_amd.householder import MatHouseholderBatch
    from fplll_amd import _lib
    h = MatHouseholderBatch(ctx, 1, 8, 8)
    h.set_basis(np.eye(8, dtype=np.int64) * 3)
    h.update_R()
    for args in ((0, 0, 0), (8, 8, 0), (5, 6, 0), (5, 2, 3), (5, 5, -1)):
        with pytest.raises(_lib.HipError):
            h.size_reduce(*args)
    red, st = h.size_reduce(5)
    assert list(red) == [0] and list(st) == [1]
    h.close()
