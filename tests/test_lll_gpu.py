"""GPU parity tests for the batched LLL kernel (fphip_gso_lll through the C ABI): the reduced
basis, the swap count, the number of zero rows and the status must equal the real reference's
(tests/golden/lll_*.json) and the C oracle's on seeded inputs — LLL is a chain of floating-point
DECISIONS, so equality of the output basis means every Lovasz test, insertion index and rounded
multiplier along the way was the reference's.  After the call mu / r must be update_gso() of the
reduced basis."""
import os

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu


def _qary(rng, d, k, q):
    """[[I_k H],[0 q I_(d-k)]] like gen_qary (nr/matrix.cpp:407-433), raw."""
    b = np.zeros((d, d), dtype=np.int64)
    b[:k, :k] = np.eye(k, dtype=np.int64)
    b[:k, k:] = rng.integers(0, q, size=(k, d - k))
    b[k:, k:] = q * np.eye(d - k, dtype=np.int64)
    return b


@pytest.mark.parametrize("path", C.lll_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_reference_fixture_parity(ctx, path):
    from fplll_amd.gso import MatGSOBatch
    f = C.load_lll_fixture(path)
    g = MatGSOBatch(ctx, 3, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * 3))
    st, info = g.lll(f["kmin"], f["kstart"], f["kend"], f["delta"], f["eta"], flags=f["flags"])
    assert list(st) == [f["status"]] * 3
    for L in range(3):
        assert info[L][1] == f["n_swaps"]
        assert info[L][2] == f["zeros"]
        assert info[L][0] == f["final_kappa"]
        assert np.array_equal(g.get_basis(L, 1)[0], f["b_out"])
    # mu / r after the call = update_gso() of the reduced basis (only meaningful without zero rows)
    if f["zeros"] == 0:
        o = C.OracleGSO(f["b_out"])
        assert o.update_all() == 1
        assert np.array_equal(g.row_expo(0), o.row_expo)
        assert np.array_equal(g.get_mu_matrix(0), o.mu)
        assert np.array_equal(g.get_r_matrix(0), o.r)
        o.close()
    g.close()


@pytest.mark.parametrize("d,n_extra", [(2, 0), (3, 0), (17, 0), (33, 1), (64, 0), (65, 0), (96, 2)])
def test_seeded_vs_oracle_heterogeneous_batch(ctx, d, n_extra):
    """A batch of DIFFERENT lattices (every wave takes its own decision path), incl. shapes at the
    chunk boundaries (64/65) and n > d."""
    from fplll_amd.gso import MatGSOBatch
    rng = np.random.default_rng(1000 + d)
    B = 6
    n = d + n_extra
    bs = []
    for L in range(B):
        if n_extra == 0 and d >= 4:
            b = _qary(rng, d, d // 2, int(rng.integers(50, 5000)))
        else:
            b = np.zeros((d, n), dtype=np.int64)
            b[:, :d] = np.eye(d, dtype=np.int64)
            b[:, d - 1 if n_extra == 0 else d:] += rng.integers(-10**6, 10**6,
                                                               size=(d, n - (d - 1 if n_extra == 0 else d)))
        bs.append(b)
    g = MatGSOBatch(ctx, B, d, n)
    g.set_basis(np.stack(bs))
    st, info = g.lll()
    for L in range(B):
        o = C.OracleGSO(bs[L])
        ost, oinfo = o.lll()
        assert st[L] == ost == 1
        assert list(info[L][:3]) == list(oinfo[:3])
        assert np.array_equal(g.get_basis(L, 1)[0], o.b)
        o.close()
    g.close()


def test_lll_is_idempotent_and_reduced(ctx):
    """size-independent properties: an LLL-reduced basis is a fixed point (0 swaps), it is
    size-reduced (|mu| <= eta) and satisfies Lovasz (delta r_{i-1} <= r_i + mu^2 r_{i-1})."""
    from fplll_amd.gso import MatGSOBatch
    rng = np.random.default_rng(7)
    d = 80
    bs = np.stack([_qary(rng, d, d // 2, 3001 + 2 * i) for i in range(4)])
    g = MatGSOBatch(ctx, 4, d, d)
    g.set_basis(bs)
    st, info = g.lll()
    assert list(st) == [1] * 4 and all(info[:, 1] > 0)
    out = g.get_basis(0, 4)
    for L in range(4):
        mu = g.get_mu_matrix(L)
        r = g.get_r_matrix(L)
        e = g.row_expo(L).astype(np.float64)
        mu_t = mu * np.exp2(e[:, None] - e[None, :])
        rd = np.diag(r) * np.exp2(2 * e)
        assert np.abs(mu_t).max() <= 0.51 + 1e-9
        lhs = 0.99 * rd[:-1]
        rhs = rd[1:] + np.diag(mu_t, -1) ** 2 * rd[:-1]
        assert np.all(lhs <= rhs * (1 + 1e-9))
        # same lattice: |det| of the q-ary basis is q^(d-k)
        assert abs(np.sum(np.log(rd)) / 2 - (d - d // 2) * np.log(3001 + 2 * L)) < 1e-6
    g.set_basis(out)
    st2, info2 = g.lll()
    assert list(st2) == [1] * 4
    assert list(info2[:, 1]) == [0] * 4
    assert np.array_equal(g.get_basis(0, 4), out)
    g.close()


def test_lll_large_batch_stress(ctx):
    """Many waves in flight (the DMA ring's vmcnt arithmetic only breaks under load)."""
    from fplll_amd.gso import MatGSOBatch
    rng = np.random.default_rng(11)
    d, B = 40, 1536
    base = [_qary(rng, d, d // 2, 1009 + 2 * i) for i in range(8)]
    bs = np.stack([base[i % 8] for i in range(B)])
    g = MatGSOBatch(ctx, B, d, d)
    g.set_basis(bs)
    st, info = g.lll()
    assert np.all(st == 1)
    out = g.get_basis(0, B)
    for i in range(8):
        o = C.OracleGSO(base[i])
        ost, oinfo = o.lll()
        assert ost == 1
        for L in range(i, B, 8):
            assert info[L][1] == oinfo[1]
            assert np.array_equal(out[L], o.b), (i, L)
        o.close()
    g.close()


def test_resident_session_equals_stateless_calls(ctx):
    """fphip_gso_session_lll: the kernel's state (rows in slots, slot table, Gram cache, mu / r, valid columns,
    verified prefix) stays on the device between calls, the caller's row operations go up as dirty rows.  Every
    call must leave what a stateless lll() of the same basis leaves: the basis, swaps / zeros, and — for the
    columns the session holds valid — the mu / r of update_gso() on that basis (they are functions of the basis).
    The row operations here are what BKZ does between lll() calls: a row replaced by a combination (insertion),
    rows rotated (move_row), a row made linearly dependent."""
    from fplll_amd.gso import MatGSOBatch
    rng = np.random.default_rng(77)
    for d in (24, 70, 130):
        b0 = _qary(rng, d, d // 2, 1009)
        g = MatGSOBatch(ctx, 1, d, d)
        ref = MatGSOBatch(ctx, 1, d, d)
        g.set_basis(b0[None])
        st, info = g.session_lll(False)
        cur = b0.copy()
        for step in range(6):
            # the stateless twin on the same input
            ref.set_basis(cur[None])
            if step == 0:
                rst, rinfo = ref.lll()
            else:
                rst, rinfo = ref.lll(0, kstart, kend)
            want = ref.get_basis(0, 1)[0]
            b, mu, r, vc, ex = g.session_read(0)
            assert int(st[0]) == int(rst[0]) == 1
            assert list(info[0][:3]) == list(rinfo[0][:3]), (d, step)
            assert np.array_equal(b, want), (d, step)
            rmu, rr = ref.get_mu_matrix(0), ref.get_r_matrix(0)
            for i in range(d):
                v = int(vc[i])
                assert 0 <= v <= i + 1
                if step == 0 or i < kend:
                    assert v == i + 1, (d, step, i, v)  # rows inside the reduced range are complete
                up = min(v, i)
                assert np.array_equal(mu[i, :up], rmu[i, :up]) and np.array_equal(r[i, :up], rr[i, :up]), (d, step, i)
                if v == i + 1:
                    assert r[i, i] == rr[i, i]
            assert np.array_equal(ex, ref.row_expo(0))
            # the caller's row operations before the next call
            cur = b.copy()
            dirty = {}
            kend = int(rng.integers(max(3, d // 2), d + 1))
            kstart = int(rng.integers(0, kend - 1))
            kind = step % 3
            if kind == 0:      # insertion: a row becomes a small combination of the rows of a block
                i = int(rng.integers(kstart, kend))
                co = rng.integers(-3, 4, size=kend - kstart)
                co[i - kstart] = 1
                cur[i] = (co[:, None] * cur[kstart:kend]).sum(axis=0)
                dirty[i] = cur[i]
            elif kind == 1:    # move_row: rotate a range
                lo = int(rng.integers(kstart, kend - 1)); hi = int(rng.integers(lo + 1, kend))
                cur[lo:hi + 1] = np.roll(cur[lo:hi + 1], 1, axis=0)
                for i in range(lo, hi + 1):
                    dirty[i] = cur[i]
            else:              # two rows changed at once, one of them by a large multiple
                i, j = (int(x) for x in rng.choice(np.arange(kstart, kend), size=2, replace=False))
                cur[i] = cur[i] + 977 * cur[j]
                dirty[i] = cur[i]
                k2 = int(rng.integers(0, kend))
                if k2 != i:
                    cur[k2] = cur[k2] - cur[i]
                    dirty[k2] = cur[k2]
            st, info = g.session_lll(True, 0, kstart, kend, dirty=dirty)
        # the other entry points refuse to run on slot-ordered rows; set_basis ends the session
        with pytest.raises(Exception):
            g.get_basis(0, 1)
        g.set_basis(cur[None])
        assert g.get_basis(0, 1)[0].shape == (d, d)
        g.close(); ref.close()


def test_siegel_in_a_session_and_a_change_of_flags(ctx):
    """LLL_SIEGEL inside a resident session equals the stateless Siegel call; switching the flag between two
    calls of one session forgets the verified prefix (it was verified against the other swap test), so the
    second call equals a stateless call with its own flags on the basis the first one left."""
    from fplll_amd.gso import MatGSOBatch
    f = C.load_lll_fixture(os.path.join(C.GOLDEN, "lll_q72_siegel.json"))
    d = f["d"]
    g = MatGSOBatch(ctx, 1, d, f["n"])
    g.set_basis(f["b_in"][None])
    st, info = g.session_lll(False, flags=4)
    b, mu, r, vc, ex = g.session_read(0)
    assert int(st[0]) == 1 and int(info[0][1]) == f["n_swaps"] and np.array_equal(b, f["b_out"])
    # Siegel-reduced is weaker than delta-LLL-reduced: a plain call on the same session has swaps left to do
    st2, info2 = g.session_lll(True, flags=0)
    b2 = g.session_read(0)[0]
    ref = MatGSOBatch(ctx, 1, d, f["n"])
    ref.set_basis(b[None])
    rst, rinfo = ref.lll()
    assert int(st2[0]) == int(rst[0]) == 1 and int(info2[0][1]) == int(rinfo[0][1]) > 0
    assert np.array_equal(b2, ref.get_basis(0, 1)[0])
    g.close(); ref.close()


@pytest.mark.parametrize("d,n_extra,flags", [(17, 0, 2), (65, 0, 2), (96, 2, 2), (40, 0, 6)])
def test_early_reduction_seeded_vs_oracle(ctx, d, n_extra, flags):
    """LLL_EARLY_RED (lll.cpp:84-99, lll.h:125-140; with LLL_SIEGEL on top in the last case) on a batch of DIFFERENT
    lattices against the C oracle's restatement (itself pinned on the reference's lll_*_earlyred fixtures): basis,
    swaps, zeros."""
    from fplll_amd.gso import MatGSOBatch
    rng = np.random.default_rng(7000 + d)
    B = 4
    n = d + n_extra
    bs = []
    for L in range(B):
        if n_extra == 0:
            b = _qary(rng, d, d // 2, int(rng.integers(50, 5000)))
        else:
            b = np.zeros((d, n), dtype=np.int64)
            b[:, :d] = np.eye(d, dtype=np.int64)
            b[:, d:] += rng.integers(-10**6, 10**6, size=(d, n - d))
        bs.append(b)
    g = MatGSOBatch(ctx, B, d, n)
    g.set_basis(np.stack(bs))
    st, info = g.lll(flags=flags)
    plain = 0
    for L in range(B):
        o = C.OracleGSO(bs[L])
        ost, oinfo = o.lll(flags=flags)
        assert st[L] == ost == 1
        assert list(info[L][:3]) == list(oinfo[:3])
        assert np.array_equal(g.get_basis(L, 1)[0], o.b)
        o.close()
        o = C.OracleGSO(bs[L])
        plain += int(o.lll(flags=flags & 4)[1][1] != oinfo[1])
        o.close()
    g.close()
    C.note(lambda: ("early reduction d=%d flags=%d: %d of %d lattices took another path than without it (swap counts)"
                    % (d, flags, plain, B),))


@pytest.mark.parametrize("name", ["lll_q40_zero1_dup2_u", "lll_q72_u", "lll_r30_earlyred_u", "lll_q72_siegel_earlyred_sub_u"])
def test_transformation_matrix_follows_the_row_operations(ctx, name):
    """MatGSO(b, u = identity, ...) (enable_transform, gso.cpp:84-158, 289-366): every row operation of the LLL run
    acts on u as well and move_row rotates its rows with b's.  The device's u equals the REAL reference's (fixtures
    of oracle/ref_driver with LLLFIX_U: a q-ary basis with a zero row and two dependent rows — where u is NOT
    determined by b_in and b_out —, a 72-dim q-ary basis, a knapsack basis under LLL_EARLY_RED), and u b_in = b_out
    in exact integers.  A non-identity start: u_out = T u_in with the same T."""
    from fplll_amd.gso import MatGSOBatch
    f = C.load_lll_fixture(os.path.join(C.GOLDEN, name + ".json"))
    d, n = f["d"], f["n"]
    g = MatGSOBatch(ctx, 2, d, n)
    g.set_basis(np.stack([f["b_in"]] * 2))
    rng = np.random.default_rng(5)
    u0 = np.stack([np.eye(d, dtype=np.int64), np.triu(rng.integers(-3, 4, size=(d, d)), 1) + np.eye(d, dtype=np.int64)])
    g.enable_transform(u0)
    st, info = g.lll(f["kmin"], f["kstart"], f["kend"], f["delta"], f["eta"], flags=f["flags"])
    assert list(st) == [f["status"]] * 2 and int(info[0][1]) == f["n_swaps"] and int(info[0][2]) == f["zeros"]
    u = g.get_transform()
    assert np.array_equal(g.get_basis(0, 1)[0], f["b_out"])
    assert np.array_equal(u[0], f["u_out"])
    ui, bi = u[0].astype(object), f["b_in"].astype(object)
    assert np.array_equal(ui.dot(bi), f["b_out"].astype(object))
    assert np.array_equal(u[1].astype(object), f["u_out"].astype(object).dot(u0[1].astype(object)))
    # the entry points that would leave u behind refuse to run
    with pytest.raises(Exception):
        g.size_reduction()
    with pytest.raises(Exception):
        g.bkz(10, f["delta"], f["eta"], 1)
    # a second call keeps accumulating: LLL of a reduced basis changes nothing, u stays
    st2, info2 = g.lll(f["kmin"], f["kstart"], f["kend"], f["delta"], f["eta"], flags=f["flags"] & 4)
    if f["zeros"] == 0:
        assert int(info2[0][1]) == 0 and np.array_equal(g.get_transform(0, 1)[0], f["u_out"])
    g.close()


@pytest.mark.parametrize("name", ["lll_q40_u_uinv", "lll_r30_u_uinv"])
def test_inverse_transformation_from_the_device_transform(ctx, name):
    """u_inv_t (enable_inverse_transform) of the reference runs that track it: MatGSOBatch.get_inverse_transform_t —
    the exact inverse transpose of the u the device tracked, taken on the host — equals the reference's matrix."""
    import json
    from fplll_amd.gso import MatGSOBatch
    path = os.path.join(C.GOLDEN, name + ".json")
    f = C.load_lll_fixture(path)
    with open(path) as fh:
        want = np.array(json.load(fh)["u_inv_t_out"], dtype=object).reshape(f["d"], f["d"])
    g = MatGSOBatch(ctx, 2, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * 2))
    g.enable_transform(None)
    st, info = g.lll(f["kmin"], f["kstart"], f["kend"], f["delta"], f["eta"], flags=f["flags"])
    assert list(st) == [f["status"]] * 2
    assert np.array_equal(g.get_transform(0, 1)[0], f["u_out"])
    uinv = g.get_inverse_transform_t()
    assert np.array_equal(uinv[0], want) and np.array_equal(uinv[1], want)
    g.close()


def test_transformation_matrix_in_a_session(ctx):
    """u through a resident session: the caller's row operations between two calls go up as dirty rows of b AND u
    (n + d integers), the session's calls keep both in step.  After every call u b_in = b in exact integers and b
    equals the stateless twin's; set_basis ends the session and leaves u in position order on the device."""
    from fplll_amd.gso import MatGSOBatch
    rng = np.random.default_rng(31)
    d = 70
    b0 = _qary(rng, d, d // 2, 3001)
    g = MatGSOBatch(ctx, 1, d, d)
    ref = MatGSOBatch(ctx, 1, d, d)
    g.set_basis(b0[None])
    g.enable_transform()
    st, info = g.session_lll(False)
    cur_u = None
    for step in range(4):
        b = g.session_read(0)[0]
        u = g.session_read_transform(0)
        assert int(st[0]) == 1
        assert np.array_equal(u.astype(object).dot(b0.astype(object)), b.astype(object)), step
        if step > 0:
            ref.set_basis(cur[None])
            rst, _ = ref.lll(0, kstart, kend)
            assert int(rst[0]) == 1 and np.array_equal(ref.get_basis(0, 1)[0], b), step
        # the caller's row operations: on b and on u alike
        cur, cur_u = b.copy(), u.copy()
        kend = int(rng.integers(d // 2, d + 1)); kstart = int(rng.integers(0, kend - 2))
        dirty = {}
        i, j = (int(x) for x in rng.choice(np.arange(kstart, kend), size=2, replace=False))
        cur[i] += 13 * cur[j]; cur_u[i] += 13 * cur_u[j]
        dirty[i] = np.concatenate([cur[i], cur_u[i]])
        lo = int(rng.integers(kstart, kend - 1)); hi = int(rng.integers(lo + 1, kend))
        cur[lo:hi + 1] = np.roll(cur[lo:hi + 1], 1, axis=0); cur_u[lo:hi + 1] = np.roll(cur_u[lo:hi + 1], 1, axis=0)
        for t in range(lo, hi + 1):
            dirty[t] = np.concatenate([cur[t], cur_u[t]])
        if i not in dirty:
            dirty[i] = np.concatenate([cur[i], cur_u[i]])
        st, info = g.session_lll(True, 0, kstart, kend, dirty=dirty)
    b = g.session_read(0)[0]
    u = g.session_read_transform(0)
    g.set_basis(b[None])  # ends the session: u comes back in position order
    assert np.array_equal(g.get_transform(0, 1)[0], u)
    assert np.array_equal(u.astype(object).dot(b0.astype(object)), b.astype(object))
    g.close(); ref.close()


def test_early_reduction_in_a_session(ctx):
    """A session is one LLLReduction object: last_early_red (lll.h:70) starts at 0 and is kept.  The first call
    equals the reference's fixture; a second call on the reduced basis finds every power of two already done
    (kappa > last_early_red fails) and, the rows being a fixed point, changes nothing."""
    from fplll_amd.gso import MatGSOBatch
    f = C.load_lll_fixture(os.path.join(C.GOLDEN, "lll_q72_earlyred.json"))
    g = MatGSOBatch(ctx, 1, f["d"], f["n"])
    g.set_basis(f["b_in"][None])
    st, info = g.session_lll(False, flags=2)
    b = g.session_read(0)[0]
    assert int(st[0]) == 1 and int(info[0][1]) == f["n_swaps"] and np.array_equal(b, f["b_out"])
    st2, info2 = g.session_lll(True, flags=2)
    assert int(st2[0]) == 1 and int(info2[0][1]) == 0 and np.array_equal(g.session_read(0)[0], f["b_out"])
    g.close()
