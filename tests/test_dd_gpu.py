"""Double-double on the device (BASELINE config 5 as stated: "HLLL (Householder, dd_real)").

The reference's FP_NR<dd_real> sits on libqd, an un-vendored optional dependency that is absent here
(SURVEY.md 8(c)): bit-for-bit parity with it is UNPINNED.  What pins the device path instead:
  1. the arithmetic of csrc/ftx.h against multiprecision (mpmath) at double-double accuracy;
  2. the Householder R-factor computed in double-double against the REAL reference run with
     FP_NR<mpfr_t> at 106 bits (tests/golden/hhmp106_*.json.gz, oracle/ref_driver.cpp `hhmp`);
  3. HLLL in double-double: status, a reduced basis of the same lattice, and — where the reference's
     own double / long double / 106-bit runs all agree — the reference's basis
     (config 5's 256-dim lattice: tests/test_a_configs_at_size_gpu.py)."""
import ctypes
import gzip
import json
import os
import time

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu
mp = pytest.importorskip("mpmath")


def _dd_op(ctx, op, a, b):
    import fplll_amd
    lib = fplll_amd.load()
    lib.fphip_debug_dd_op.restype = ctypes.c_int
    n = a.shape[0]
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in (a[:, 0], a[:, 1], b[:, 0], b[:, 1])]
    ohi, olo = np.empty(n), np.empty(n)
    vp = ctypes.c_void_p
    rc = lib.fphip_debug_dd_op(ctx.handle, op, n, *[vp(x.ctypes.data) for x in arrs],
                               vp(ohi.ctypes.data), vp(olo.ctypes.data))
    assert rc == 0
    return ohi, olo


def _rand_dd(rng, n, lo_exp=-30, hi_exp=30):
    hi = rng.standard_normal(n) * np.exp2(rng.integers(lo_exp, hi_exp, n).astype(np.float64))
    lo = hi * np.exp2(-53.0) * rng.uniform(-0.5, 0.5, n)
    s = hi + lo  # renormalise: |lo| <= ulp(hi)/2
    return np.stack([s, lo - (s - hi)], axis=1)


def test_double_double_arithmetic_against_mpmath(ctx):
    mp.mp.prec = 400
    rng = np.random.default_rng(5)
    n = 2000
    a, b = _rand_dd(rng, n), _rand_dd(rng, n)
    b[:, 0] = np.where(b[:, 0] == 0, 1.0, b[:, 0])
    val = lambda x: mp.mpf(float(x[0])) + mp.mpf(float(x[1]))  # noqa: E731
    eps = mp.mpf(2) ** -104
    for op, name, fn, tol in ((0, "add", lambda x, y: x + y, 1), (1, "sub", lambda x, y: x - y, 1),
                              (2, "mul", lambda x, y: x * y, 4), (3, "div", lambda x, y: x / y, 8),
                              (4, "sqrt", lambda x, y: mp.sqrt(abs(x)), 4)):
        aa = np.abs(a) if op == 4 else a
        if op == 4:
            aa = np.stack([np.abs(a[:, 0]), np.where(a[:, 0] < 0, -a[:, 1], a[:, 1])], axis=1)
        ohi, olo = _dd_op(ctx, op, aa, b)
        worst = mp.mpf(0)
        for i in range(n):
            x, y = val(aa[i]), val(b[i])
            want = fn(x, y)
            got = mp.mpf(float(ohi[i])) + mp.mpf(float(olo[i]))
            # add / sub ("sloppy" libqd addition): the error is relative to the larger operand
            scale = max(abs(x), abs(y)) if op < 2 else abs(want)
            err = abs(got - want) / scale if scale != 0 else abs(got - want)
            worst = max(worst, err)
        assert worst <= tol * eps, (name, mp.nstr(worst, 5))
        C.note(lambda: ("dd %s: worst error %s (in units of 2^-104)" % (name, mp.nstr(worst / eps, 4)),))
    # nint: exact
    q = _rand_dd(rng, n, 0, 60)
    q[::7, 0] = np.round(q[::7, 0])  # integral high words: the low word decides
    q[1::7, 0] = np.floor(q[1::7, 0]) + 0.5  # ties of the high word: the low word breaks them
    q[:, 1] = np.where(np.abs(q[:, 0]) < 2 ** 52, q[:, 1], np.round(q[:, 1]) + 0.25)
    s = q[:, 0] + q[:, 1]
    q = np.stack([s, q[:, 1] - (s - q[:, 0])], axis=1)
    ohi, olo = _dd_op(ctx, 5, q, q)
    for i in range(n):
        x = val(q[i])
        got = mp.mpf(float(ohi[i])) + mp.mpf(float(olo[i]))
        assert got == mp.floor(got) and abs(got - x) <= mp.mpf(1) / 2, (q[i], ohi[i], olo[i])


@pytest.mark.parametrize("name", ["q40", "q72"])
def test_r_factor_in_double_double_against_mpfr106(ctx, name):
    """HLLL in double-double on an already HLLL-reduced basis leaves it alone and ends with the
    R-factor of that basis in R: compared entry by entry with the reference's MatHouseholder run in
    MPFR at 106 bits.  (The plain-double kernel on the same input agrees to ~1e-13 only.)"""
    from fplll_amd.householder import MatHouseholderBatch
    mp.mp.prec = 300
    with gzip.open(os.path.join(C.GOLDEN, "hhmp106_%s.json.gz" % name), "rt") as f:
        j = json.load(f)
    d, n = j["d"], j["n"]
    b = np.array(j["b"], dtype=np.int64).reshape(d, n)
    want = iter(j["R"])
    h = MatHouseholderBatch(ctx, 2, d, n, row_expo=True)
    worst = {}
    for prec in (106, 53):
        h.set_basis(np.stack([b] * 2))
        st, info = h.hlll(precision=prec)
        assert list(st) == [1, 1] and int(info[0][0]) == 0  # no swap: the basis was reduced already
        assert np.array_equal(h.get_basis(0, 1)[0], b)
        R, e = h.get_R(1)
        Rlo = h.get_R_lo(1) if prec == 106 else np.zeros_like(R)
        w = mp.mpf(0)
        want = iter(j["R"])
        for i in range(d):
            rown = mp.mpf(0)
            row = []
            for jj in range(i + 1):
                ref = mp.mpf(next(want))
                got = (mp.mpf(float(R[i, jj])) + mp.mpf(float(Rlo[i, jj]))) * mp.mpf(2) ** int(e[i])
                row.append((ref, got))
                rown += ref * ref
            rown = mp.sqrt(rown)
            for ref, got in row:
                w = max(w, abs(got - ref) / rown)
        worst[prec] = w
    C.note(lambda: ("R-factor %s (%dx%d) vs MPFR-106: double-double %s, double %s (relative to the row norm)"
          % (name, d, n, mp.nstr(worst[106], 4), mp.nstr(worst[53], 4)),))
    assert worst[106] <= mp.mpf(2) ** -92   # ~1e-28: double-double accuracy with d*n roundings of slack
    assert worst[53] <= mp.mpf(2) ** -40
    assert worst[106] * 2 ** 30 < worst[53] or worst[53] == 0
    h.close()


def _same_lattice(b_in, b_out):
    """rows of b_out generate the lattice of b_in: b_out = U b_in with U integral and |det U| = 1
    (exact rational elimination)."""
    from fractions import Fraction
    d, n = b_in.shape
    assert d == n, "square bases only"
    # solve X b_in = b_out  <=>  b_in^T X^T = b_out^T : Gauss-Jordan on [b_in^T | b_out^T]
    A = [[Fraction(int(b_in[j, i])) for j in range(d)] + [Fraction(int(b_out[j, i])) for j in range(d)]
         for i in range(n)]
    det = Fraction(1)
    for c in range(d):
        piv = next(r for r in range(c, n) if A[r][c] != 0)
        if piv != c:
            A[c], A[piv] = A[piv], A[c]
            det = -det
        det *= A[c][c]
        inv = 1 / A[c][c]
        A[c] = [v * inv for v in A[c]]
        for r in range(n):
            if r != c and A[r][c] != 0:
                f = A[r][c]
                A[r] = [a - f * p for a, p in zip(A[r], A[c])]
    X = [[A[i][d + j] for i in range(d)] for j in range(d)]  # X[j][i]
    if any(v.denominator != 1 for row in X for v in row):
        return False
    # |det X| = |det b_out| / |det b_in| must be 1: compare absolute determinants via the same elimination
    import sympy
    return abs(sympy.Matrix([[int(v) for v in row] for row in X]).det()) == 1


def _reference_says_hlll_reduced(b):
    """The reference's own predicate (is_hlll_reduced, hlll.cpp:507-585, the assertion of its
    tests/test_hlll.cpp), evaluated by the reference at 212 bits of MPFR (ref_driver ishlll)."""
    import subprocess
    import tempfile
    drv = os.path.join(C.ROOT, "oracle", "_ref", "ref_driver")
    assert os.path.exists(drv), "oracle/_ref/ref_driver is not built"
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as t:
        t.write("[" + "\n".join("[" + " ".join(str(int(x)) for x in row) + "]" for row in b) + "]\n")
    try:
        r = subprocess.run([drv, "ishlll", t.name, "212"], capture_output=True, text=True, timeout=600)
        return json.loads(r.stdout)["reduced"] == 1
    finally:
        os.unlink(t.name)


@pytest.mark.parametrize("name", ["hlll_q40", "hlll_r30", "hlll_u24"])
def test_hlll_in_quad_double_on_reference_fixtures(ctx, name):
    """hlll(precision=212): the reference's algorithm in QUAD-double arithmetic on the device (ftx.h QD, the stand-in
    for FP_NR<qd_real>: the third stage of the wrapper's ladder, wrapper.cpp:630-710) — success, the reference's
    output basis (which fplll returns in double, long double and MPFR alike on these inputs) and the swap count of
    the double-double run."""
    from fplll_amd.householder import MatHouseholderBatch
    f = C.load_hlll_fixture(os.path.join(C.GOLDEN, name + ".json"))
    h = MatHouseholderBatch(ctx, 2, f["d"], f["n"], row_expo=True)
    swaps = {}
    for prec in (106, 212):
        h.set_basis(np.stack([f["b_in"]] * 2))
        st, info = h.hlll(f["delta"], f["eta"], f["theta"], f["c"], precision=prec)
        assert list(st) == [1, 1]
        out = h.get_basis(0, 2)
        assert np.array_equal(out[0], f["b_out"]) and np.array_equal(out[1], f["b_out"])
        swaps[prec] = int(info[0][0])
        C.note(lambda: ("%s precision %d: %d swaps, %.1f ms" % (name, prec, swaps[prec], h.last_kernel_ms),))
    assert swaps[106] == swaps[212]
    h.close()


def test_precision_ladder_reaches_quad_double(ctx, monkeypatch):
    """fphip_hh_hlll_ladder with its third stage: FPHIP_HLLL_LADDER_TEST=2 pretends the odd lattices failed in
    double and every fourth also in double-double, so that they go on in quad-double from the basis the stage before
    left — every lattice ends on the reference's basis, the stages are 53 / 106 / 53 / 212 / ..."""
    from fplll_amd.householder import MatHouseholderBatch
    f = C.load_hlll_fixture(os.path.join(C.GOLDEN, "hlll_q40.json"))
    monkeypatch.setenv("FPHIP_HLLL_LADDER_TEST", "2")
    h = MatHouseholderBatch(ctx, 8, f["d"], f["n"], row_expo=True)
    h.set_basis(np.stack([f["b_in"]] * 8))
    st, info, stage = h.hlll_ladder(f["delta"], f["eta"], f["theta"], f["c"])
    assert list(st) == [1] * 8
    assert list(stage) == [53, 106, 53, 212, 53, 106, 53, 212]
    out = h.get_basis(0, 8)
    assert all(np.array_equal(out[L], f["b_out"]) for L in range(8))
    h.close()


@pytest.mark.parametrize("name", ["hlll_q40", "hlll_q72"])
def test_blocked_reflector_application_in_hlll(ctx, name, monkeypatch):
    """FPHIP_HLLL_BLOCKED=1: update_R inside the HLLL loop in compact-WY form — the reflectors sixteen at a time with
    a T per block kept current by update_R_last (hlll_x.hip: m independent dot products in one butterfly, a 16 x 16
    triangle, m AXPYs).  Another order of the sums than one by one, the same decisions: the reference's output basis
    and the swap count of the one-by-one mode, in double-double and in double.  (Opt-in: on config 5's lone wave it
    is the slower of the two — the A/B is in DESIGN.md section 4d.)"""
    from fplll_amd.householder import MatHouseholderBatch
    path = os.path.join(C.GOLDEN, name + ".json")
    if not os.path.exists(path):
        pytest.skip("fixture absent")
    f = C.load_hlll_fixture(path)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("FPHIP_HLLL_BLOCKED", mode)
        h = MatHouseholderBatch(ctx, 3, f["d"], f["n"], row_expo=True)
        for prec in (106, 53):
            h.set_basis(np.stack([f["b_in"]] * 3))
            st, info = h.hlll(f["delta"], f["eta"], f["theta"], f["c"], precision=prec)
            assert list(st) == [1, 1, 1]
            res[(mode, prec)] = (h.get_basis(0, 3), [int(x[0]) for x in info])
        h.close()
    for prec in (106, 53):
        (b0, s0), (b1, s1) = res[("0", prec)], res[("1", prec)]
        assert all(np.array_equal(b1[L], f["b_out"]) for L in range(3)) or prec == 53
        assert all(np.array_equal(b1[L], b0[L]) for L in range(3))
        assert s0 == s1


@pytest.mark.parametrize("path", C.hlll_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_hlll_in_double_double_on_reference_fixtures(ctx, path):
    """hlll(precision=106) on the inputs of the reference fixtures: success, and the reference's
    output basis for the q-ary / knapsack / uniform lattices (far from any tie: every decision of the
    algorithm has tens of bits of margin, and the reference returns the same basis at 106 bits).
    The NTRU-like hlll_n64 is different by nature: the rotations of a vector have EXACTLY equal norms,
    so Lovasz comparisons tie in exact arithmetic and rounding noise decides them — the reference's
    own double and 106-bit MPFR runs return different bases (tests/golden/hlllmp106_n64.json), and
    double-double is a third arithmetic.  There the check is the reference's own acceptance test
    (tests/test_hlll.cpp): its predicate is_hlll_reduced, evaluated BY the reference at 212 bits, on
    the device's output, plus: same lattice (exact)."""
    from fplll_amd.householder import MatHouseholderBatch
    f = C.load_hlll_fixture(path)
    ntru = "n64" in path
    h = MatHouseholderBatch(ctx, 2, f["d"], f["n"], row_expo=True)
    for prec in (106, 53):
        h.set_basis(np.stack([f["b_in"]] * 2))
        st, info = h.hlll(f["delta"], f["eta"], f["theta"], f["c"], precision=prec)
        assert list(st) == [1, 1]
        out = h.get_basis(0, 2)
        assert np.array_equal(out[0], out[1])
        same = np.array_equal(out[0], f["b_out"])
        C.note(lambda: ("%s precision %d: %d swaps, %.1f ms, output %s the reference's" %
              (os.path.basename(path), prec, int(info[0][0]), h.last_kernel_ms, "==" if same else "!="),))
        if ntru:
            assert _reference_says_hlll_reduced(out[0]) and _same_lattice(f["b_in"], out[0])
        elif prec == 106:
            assert same
    h.close()


def test_precision_ladder_double_then_double_double(ctx, monkeypatch):
    """fphip_hh_hlll_ladder — the wrapper's ladder (wrapper.cpp:478-529) with both stages on the
    device.  On these inputs the double stage succeeds (stage 53 everywhere, results of hlll());
    with FPHIP_HLLL_LADDER_TEST=1 the odd lattices are sent on to the double-double stage as if the
    double stage had raised a precision alarm: they continue from the basis stage 1 left, end with
    status 1 at stage 106 and the same basis."""
    from fplll_amd.householder import MatHouseholderBatch
    f = C.load_hlll_fixture(os.path.join(C.GOLDEN, "hlll_q40.json"))
    h = MatHouseholderBatch(ctx, 4, f["d"], f["n"], row_expo=True)
    h.set_basis(np.stack([f["b_in"]] * 4))
    st, info, stage = h.hlll_ladder(f["delta"], f["eta"], f["theta"], f["c"])
    assert list(st) == [1] * 4 and list(stage) == [53] * 4
    assert all(np.array_equal(b, f["b_out"]) for b in h.get_basis(0, 4))
    monkeypatch.setenv("FPHIP_HLLL_LADDER_TEST", "1")
    h.set_basis(np.stack([f["b_in"]] * 4))
    st, info, stage = h.hlll_ladder(f["delta"], f["eta"], f["theta"], f["c"])
    assert list(st) == [1] * 4 and list(stage) == [53, 106, 53, 106]
    assert all(np.array_equal(b, f["b_out"]) for b in h.get_basis(0, 4))
    h.close()


# ---------------------------------------------------------------------------------------------
# LLL in double-double (lll_x.hip, fphip_gso_lll_ex) — the second stage of the LLL-side precision
# ladder (Wrapper::lll, wrapper.cpp:281-359)
# ---------------------------------------------------------------------------------------------
def _basisstat(b):
    """ref_driver basisstat: the reference's is_lll_reduced (lll.cpp:226-257) at 256 bits, volume, slope."""
    import subprocess
    import tempfile
    drv = os.path.join(C.ROOT, "oracle", "_ref", "ref_driver")
    assert os.path.exists(drv), "oracle/_ref/ref_driver is not built"
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as t:
        t.write("[" + "\n".join("[" + " ".join(str(int(x)) for x in row) + "]" for row in b) + "]\n")
    try:
        r = subprocess.run([drv, "basisstat", t.name], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-300:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    finally:
        os.unlink(t.name)


@pytest.mark.parametrize("name", ["lll_q40", "lll_q72", "lll_u24", "lll_r30", "lll_q40_zero2", "lll_q40_dup2"])
def test_lll_in_double_double_and_plain_double_tree_order(ctx, name):
    """fphip_gso_lll_ex at 106 and 53 bits on the LLL fixtures (q-ary, uniform, integer relation, with
    zero and duplicated rows): RED_SUCCESS, the output is LLL-reduced by the REFERENCE's predicate
    (is_lll_reduced, delta 0.99 / eta 0.51, evaluated at 256 bits) and spans the input's lattice; on
    these well-conditioned inputs the double-double run also returns the reference's own basis."""
    from fplll_amd.gso import MatGSOBatch
    f = C.load_lll_fixture(os.path.join(C.GOLDEN, name + ".json"))
    g = MatGSOBatch(ctx, 2, f["d"], f["n"])
    for prec in (106, 53):
        g.set_basis(np.stack([f["b_in"]] * 2))
        st, info = g.lll_ex(prec, f["kmin"], f["kstart"], f["kend"], f["delta"], f["eta"])
        out = g.get_basis(0, 2)
        assert list(st) == [1, 1], (prec, st, info)
        assert np.array_equal(out[0], out[1])
        nz = out[0][np.any(out[0] != 0, axis=1)]
        if f["kmin"] == 0 and f["kend"] in (-1, f["d"]):
            s = _basisstat(nz)
            assert s["is_lll_reduced"] == 1, (name, prec, s)
            ref = _basisstat(f["b_out"][np.any(f["b_out"] != 0, axis=1)])
            assert abs(s["log_volume"] - ref["log_volume"]) < 1e-9 * max(1.0, abs(ref["log_volume"]))
        same = np.array_equal(out[0], f["b_out"])
        C.note(lambda: ("%s at %d bits: %d swaps (reference %d), %.1f ms, basis %s the reference's"
              % (name, prec, int(info[0][1]), f["n_swaps"], g.last_kernel_ms, "==" if same else "!="),))
        if prec == 106 and name in ("lll_q40", "lll_q72", "lll_u24"):
            assert same and int(info[0][1]) == f["n_swaps"]
    g.close()


@pytest.mark.parametrize("name", ["lll_q40", "lll_u24"])
def test_lll_in_quad_double_and_the_three_stage_ladder(ctx, name, monkeypatch):
    """fphip_gso_lll_ex at 212 bits (quad-double, the stand-in for FP_NR<qd_real>: Wrapper::lll's third fast type):
    the reference's basis and swap count on well-conditioned fixtures, like the double-double run; and
    fphip_gso_lll_ladder with its third stage (FPHIP_LLL_LADDER_TEST=2 pretends the odd lattices failed in double and
    every fourth in double-double): stages 53 / 106 / 53 / 212, every lattice on the reference's basis."""
    from fplll_amd.gso import MatGSOBatch
    f = C.load_lll_fixture(os.path.join(C.GOLDEN, name + ".json"))
    g = MatGSOBatch(ctx, 2, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * 2))
    st, info = g.lll_ex(212, f["kmin"], f["kstart"], f["kend"], f["delta"], f["eta"])
    out = g.get_basis(0, 2)
    assert list(st) == [1, 1], (st, info)
    assert np.array_equal(out[0], f["b_out"]) and np.array_equal(out[1], f["b_out"])
    assert int(info[0][1]) == f["n_swaps"]
    C.note(lambda: ("%s at 212 bits: %d swaps, %.1f ms" % (name, int(info[0][1]), g.last_kernel_ms),))
    g.close()
    if f["kmin"] != 0 or f["kstart"] != 0:
        return
    monkeypatch.setenv("FPHIP_LLL_LADDER_TEST", "2")
    g = MatGSOBatch(ctx, 8, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * 8))
    st, info, stage = g.lll_ladder(f["kmin"], f["kstart"], f["kend"], f["delta"], f["eta"])
    assert list(st) == [1] * 8
    assert list(stage) == [53, 106, 53, 212, 53, 106, 53, 212]
    out = g.get_basis(0, 8)
    assert all(np.array_equal(out[L], f["b_out"]) for L in range(8))
    g.close()


def _rows_in_qary_lattice(b_in, b_out):
    """b_in = [[I, H], [0, q I]] (q-ary / NTRU-like generator): every row (x, y) of b_out lies in its
    lattice iff y = x H (mod q).  With equal volumes (basisstat) that is lattice equality."""
    d = b_in.shape[0]
    k = d // 2
    assert np.array_equal(b_in[:k, :k], np.eye(k, dtype=np.int64)) and not b_in[k:, :k].any()
    q = int(b_in[k, k])
    assert np.array_equal(b_in[k:, k:], q * np.eye(d - k, dtype=np.int64))
    H = b_in[:k, k:].astype(object)
    x, y = b_out[:, :k].astype(object), b_out[:, k:].astype(object)
    return bool(np.all((y - x.dot(H)) % q == 0))


def test_config5_lattice_lll_ladder_escalates_to_double_double(ctx):
    """A GENUINE escalation of the LLL-side precision ladder (Wrapper::lll, wrapper.cpp:281-359), on
    BASELINE config 5's own lattice (`latticegen n 128 12 b`, 256-dim NTRU-like): LLLReduction in double
    stops with RED_BABAI_FAILURE on it — the reference (`fplll -a lll -m fast -f double`: "infinite loop
    in babai") and the exact-order device kernel alike — and fphip_gso_lll_ladder carries on in
    double-double on the device (lll_x.hip) to RED_SUCCESS.  The reference has no dd_real here and its
    long double / 106-bit MPFR runs return DIFFERENT reduced bases of this lattice (exact ties), so the
    output is judged by the reference's own predicate: is_lll_reduced (delta 0.99, eta 0.51, at 256
    bits), plus lattice equality (every row in the input's lattice, same volume)."""
    from fplll_amd.gso import MatGSOBatch
    f = C.load_hlll_fixture(os.path.join(C.GOLDEN, "c5_hlll_n256_double.json.gz"))
    b = f["b_in"]
    assert b.shape == (256, 256)
    g = MatGSOBatch(ctx, 1, 256, 256)
    g.set_basis(np.stack([b]))
    t = time.time()
    st, _ = g.lll()
    t_double = time.time() - t
    assert int(st[0]) == -1, "double LLL is expected to fail with RED_BABAI_FAILURE on this lattice (%r)" % (st,)
    g.set_basis(np.stack([b]))
    t = time.time()
    st, info, stage = g.lll_ladder()
    t_ladder = time.time() - t
    out = g.get_basis(0, 1)[0]
    g.close()
    assert int(st[0]) == 1 and int(stage[0]) == 106, (st, stage, info)
    s_out, s_in = _basisstat(out), _basisstat(b)
    C.note(lambda: ("config 5 LLL: double stops with RED_BABAI_FAILURE after %.1f s; ladder (double -> double-double) %.1f s, "
          "%d swaps in all; is_lll_reduced %d, slope %.6f (input %.6f)"
          % (t_double, t_ladder, int(info[0][1]), s_out["is_lll_reduced"], s_out["slope"], s_in["slope"]),))
    assert s_out["is_lll_reduced"] == 1 and s_in["is_lll_reduced"] == 0
    assert abs(s_out["log_volume"] - s_in["log_volume"]) < 1e-9 * abs(s_in["log_volume"])
    assert _rows_in_qary_lattice(b, out)
