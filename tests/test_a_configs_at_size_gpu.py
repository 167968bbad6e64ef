"""BASELINE.json's configurations AT THEIR FULL SIZE on the device, through the C ABI, against golden
vectors of the real reference (tests/golden/c2_*, c3_*, c5_*; the CPU suite pins the oracle to the
same files):

  config 2  BKZ-20 (BKZ_DEFAULT, no strategies) to convergence on the 120-dim q-ary lattice
            (fplll/bkz.cpp:522-672): basis, status, 138 tours' worth of work, 10 252 068 nodes
  config 3  the beta = 60 blocks of the 180-dim q-ary lattice under the pruner's coefficients
            (enumeration plugin, fplll/enum/enumerate_ext.cpp:48-167) and ONE BKZ-60 tour with the
            pruner strategies on the device (fplll/bkz.cpp:274-399): basis, status, 1 224 293 770 nodes
  config 5  HLLL on the 256-dim NTRU-like lattice, FT = double (fplll/hlll.cpp:26-173): basis, status,
            146 491 swaps — the NQ = 4 instantiation (more than 192 columns) of the HLLL kernel

plus one NQ = 4 case (d = n = 256) for each of the sweep, LLL and Householder kernels."""
import os
import time

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu


def _nodes(info_row):
    return (int(info_row[1]) & 0xffffffff) | ((int(info_row[2]) & 0xffffffff) << 32)


def _c5():
    return C.load_hlll_fixture(os.path.join(C.GOLDEN, "c5_hlll_n256_double.json.gz"))


def _q200():
    """200-dim LLL-reduced q-ary basis (reference `fplll -a lll`, make_fixtures.sh): more than 192
    columns = the NQ = 4 instantiation of every kernel.  (The GSO of config 5's 256-dim NTRU-like
    basis is beyond plain doubles — babai fails on it in the reference as well — hence this one.)"""
    from fplll_amd.gso import load_basis_txt
    return load_basis_txt(os.path.join(C.GOLDEN, "basis_q200_seed7_lll.txt.gz"))


# ---------------------------------------------------------------------------------------------
# The at-size runs that are ONE launch keeping one wavefront busy for minutes (the reduction kernels are
# throughput devices) live in tests/test_at_size_long_runs.py under the marker `gpu_long`: they are not part of
# `-m gpu` (round 5 ran them as host threads beside the other 260 tests: 625 s of a 1200 s limit, and a suite whose
# colour depended on how the box scheduled three contexts).  The helpers stay here.
# ---------------------------------------------------------------------------------------------
def _run_config3_tour(out):
    """One BKZ-60 tour (BKZ_MAX_LOOPS 1, BKZ_GH_BND) of the 180-dim lattice with the pruner
    strategies, on the device: 15 160 enumerations, 11 rerandomisations, 1.22e9 nodes."""
    import fplll_amd
    from fplll_amd.gso import MatGSOBatch
    try:
        f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "c3_bkz60_tour_strategies.json.gz"))
        ctx3 = fplll_amd.Context(int(os.environ.get("LOCAL_RANK", "0")), priority=-1)
        B = 2
        g = MatGSOBatch(ctx3, B, f["d"], f["n"])
        g.set_basis(np.stack([f["b_in"]] * B))
        rnd, draws = C.gmp_streams_native(B, f["rng_seed"])
        t = time.time()
        st, info = g.bkz_strategies(f["block_size"], f["strategies"], rnd, f["delta"], f["eta"],
                                    max_loops=f["max_loops"], gh_bnd=True, gh_factor=f["gh_factor"])
        out["c3"] = dict(wall=time.time() - t, st=[int(x) for x in st], nodes=[_nodes(i) for i in info],
                         basis_ok=[bool(np.array_equal(b, f["b_out"])) for b in g.get_basis()],
                         expect=(f["status"], f["nodes"]), ref_s=f["ref_seconds"])
        g.close()
        ctx3.close()
    except BaseException as e:  # noqa: reported by the joining test
        out["c3_error"] = repr(e)


def _basisstat(b):
    """The reference's own acceptance predicate on a basis (oracle/_ref/ref_driver basisstat:
    is_lll_reduced at 256 bits, lll.cpp:226-257; slope of log r_ii, gso_interface.cpp:198-218; volume)."""
    import json
    import subprocess
    import tempfile
    drv = os.path.join(C.ROOT, "oracle", "_ref", "ref_driver")
    assert os.path.exists(drv), "oracle/_ref/ref_driver is missing"
    f = tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False)
    f.write("[" + "\n".join("[" + " ".join(str(int(x)) for x in row) + "]" for row in b) + "]\n")
    f.close()
    try:
        r = subprocess.run([drv, "basisstat", f.name], capture_output=True, text=True, timeout=600)
    finally:
        os.unlink(f.name)
    assert r.returncode == 0, r.stderr[-500:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def _run_config3_tour_handoff(out):
    """The same tour in hand-off mode (FPHIP_BKZ_HANDOFF): the large blocks — 240 of the 15 160
    enumerations, 95 % of the nodes — go to the multi-wave enumerator, everything else stays with the
    lattice's wave.  A pruned enumeration with a shrinking radius is order dependent, so the acceptance
    test is the reference's reducedness predicate on the output, not the golden basis."""
    import fplll_amd
    from fplll_amd.gso import MatGSOBatch
    try:
        f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "c3_bkz60_tour_strategies.json.gz"))
        ctx4 = fplll_amd.Context(int(os.environ.get("LOCAL_RANK", "0")), priority=-1)
        g = MatGSOBatch(ctx4, 1, f["d"], f["n"])
        g.set_basis(np.stack([f["b_in"]]))
        rnd, draws = C.gmp_streams_native(1, f["rng_seed"])
        t = time.time()
        st, info = g.bkz_strategies(f["block_size"], f["strategies"], rnd, f["delta"], f["eta"],
                                    max_loops=f["max_loops"], gh_bnd=True, gh_factor=f["gh_factor"], handoff=True)
        wall = time.time() - t
        b = g.get_basis()[0]
        out["c3h"] = dict(wall=wall, st=int(st[0]), nodes=_nodes(info[0]), calls=int(info[0][3]),
                          stat=_basisstat(b), ref_stat=_basisstat(f["b_out"]), in_stat=_basisstat(f["b_in"]),
                          expect_status=f["status"], ref_s=f["ref_seconds"], ref_nodes=f["nodes"])
        g.close()
        ctx4.close()
    except BaseException as e:  # noqa: reported by the joining test
        out["c3h_error"] = repr(e)


def _run_config5_hlll(out):
    import fplll_amd
    from fplll_amd.householder import MatHouseholderBatch
    try:
        f = _c5()
        assert (f["d"], f["n"]) == (256, 256)
        ctx2 = fplll_amd.Context(int(os.environ.get("LOCAL_RANK", "0")), priority=-1)
        B = 2
        h = MatHouseholderBatch(ctx2, B, 256, 256, row_expo=True)
        h.set_basis(np.stack([f["b_in"]] * B))
        t = time.time()
        st, info = h.hlll(f["delta"], f["eta"], f["theta"], f["c"])
        out["c5"] = dict(wall=time.time() - t, st=[int(x) for x in st], swaps=[int(i[0]) for i in info],
                         basis_ok=[bool(np.array_equal(b, f["b_out"])) for b in h.get_basis(0, B)],
                         expect=f["status"], ref_s=f["ref_seconds"])
        h.close()
        ctx2.close()
    except BaseException as e:  # noqa: reported by the joining test
        out["c5_error"] = repr(e)


# ---------------------------------------------------------------------------------------------
# NQ = 4 (193..256 rows / columns): four registers per lane in every kernel
# ---------------------------------------------------------------------------------------------
def test_nq4_size_reduction_sweep_matches_oracle(ctx):
    """200x200, every mu pushed out of [-1/2, 1/2] (a multiplier for every j < i, different for
    each lattice of the batch): b, stored mu, r, row exponents bit-identical to the oracle's
    size_reduction(0, 200)."""
    from fplll_amd.gso import MatGSOBatch, _dense_unreduced, _unreduced_copy
    base = _q200()
    bs = [_dense_unreduced(base, 3, 11), _dense_unreduced(base, 2, 12), _unreduced_copy(base, 3, 13)]
    g = MatGSOBatch(ctx, 3, 200, 200)
    g.set_basis(np.stack(bs))
    st = g.size_reduction()
    assert list(st) == [1, 1, 1]
    for L in range(3):
        o = C.OracleGSO(bs[L])
        assert o.size_reduction(0, 200) == 1
        assert np.array_equal(g.get_basis(L, 1)[0], o.b)
        assert np.array_equal(g.get_mu_matrix(L), o.mu)
        assert np.array_equal(g.get_r_matrix(L), o.r)
        assert np.array_equal(g.row_expo(L), o.row_expo)
        o.close()
    g.close()


def test_nq4_lll_matches_oracle(ctx):
    """LLLReduction::lll on a 200x200 basis that needs a few hundred swaps (the reduced basis with
    perturbed and exchanged rows): reduced basis, status, swap count identical to the oracle."""
    from fplll_amd.gso import MatGSOBatch, _unreduced_copy
    b = _unreduced_copy(_q200(), 2, 5)
    rng = np.random.default_rng(3)
    for _ in range(24):  # exchange neighbouring rows: Lovasz failures for the loop to repair
        i = int(rng.integers(0, 199))
        b[[i, i + 1]] = b[[i + 1, i]]
    o = C.OracleGSO(b)
    st_o, info_o = o.lll()
    g = MatGSOBatch(ctx, 2, 200, 200)
    g.set_basis(np.stack([b] * 2))
    st, info = g.lll()
    assert list(st) == [st_o, st_o] and st_o == 1
    assert int(info[0][1]) == int(info_o[1]) > 0, (info[0], info_o)
    for L in range(2):
        assert np.array_equal(g.get_basis(L, 1)[0], o.b)
    C.note(lambda: ("NQ=4 LLL: %d swaps, kernel %.1f ms" % (int(info[0][1]), g.last_kernel_ms),))
    o.close()
    g.close()


def test_nq4_householder_matches_oracle(ctx):
    from fplll_amd.householder import MatHouseholderBatch
    from fplll_amd.gso import _unreduced_copy
    b = _c5()["b_out"] + 0  # config 5's reduced basis (256 columns)
    Ro, Vo, so, eo = C.oracle_hh_update_all(b, True)
    h = MatHouseholderBatch(ctx, 3, 256, 256, row_expo=True)
    h.set_basis(np.stack([b] * 3))
    assert list(h.update_R()) == [1, 1, 1]
    for L in (0, 2):
        R, e = h.get_R(L)
        assert np.array_equal(e, eo)
        assert np.array_equal(np.tril(R[:, :256]), np.tril(Ro[:, :256]))
    h.close()


# ---------------------------------------------------------------------------------------------
# config 2
# ---------------------------------------------------------------------------------------------
def test_config2_bkz20_q120_matches_reference(ctx):
    from fplll_amd.gso import MatGSOBatch
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "c2_bkz20_q120.json.gz"))
    assert (f["d"], f["block_size"], f["max_loops"]) == (120, 20, 0)
    B = 2
    g = MatGSOBatch(ctx, B, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * B))
    t = time.time()
    st, _ = g.lll()  # the fixture's input is LLL-reduced: the device confirms it (bkz.cpp:537)
    assert list(st) == [1] * B
    st, info = g.bkz(f["block_size"], f["delta"], f["eta"], f["max_loops"])
    wall = time.time() - t
    out = g.get_basis()
    C.note(lambda: ("config 2: %d tours, %d nodes, %.1f s on the device (reference %.2f s on one core)"
          % (int(info[0][0]), _nodes(info[0]), wall, f["ref_seconds"]),))
    for L in range(B):
        assert st[L] == f["status"] == 1
        assert _nodes(info[L]) == f["nodes"] == 10252068
        assert np.array_equal(out[L], f["b_out"])
    g.close()


# ---------------------------------------------------------------------------------------------
# config 3
# ---------------------------------------------------------------------------------------------
def _check_pruned_result(f, ev, res):
    """Parity of a pruned enumeration whose radius shrinks (enumerate.cpp:231-239) is order
    dependent by nature: whichever candidate is met first shrinks every level bound
    pruning[k]*radius, and may thereby cut the branch of a candidate another order would have met —
    the reference's sequential walk and a parallel walk may legitimately end on different vectors
    (enumlib has the same property, README.md:315).  What IS checkable, and checked here:
      1. the device's vector has the squared norm it reported (recomputed from mu, r);
      2. completeness: a sequential walk (the oracle) started at the device's final radius finds
         nothing shorter — every node inside the pruned region of that radius was inside the
         device's region all along, so a shorter vector there would be one the device missed;
      3. the same holds from the reference's side: its result is not shorter than the device's,
         unless its vector lies outside the pruned region of the device's final radius."""
    from fplll_amd.enumeration import FastEvaluator
    dist, x = ev.solutions[0][0], np.array(ev.solutions[0][1], dtype=np.float64)
    d = f["d"]
    # |sum_i x_i b*_i ...|^2 = sum_k r_k (x_k + sum_{j>k} mu_jk x_j)^2, mut[k][j] = mu(j,k)
    tot = 0.0
    for k in range(d):
        c = x[k] + float(np.dot(f["mut"][k, k + 1:], x[k + 1:]))
        tot += f["rdiag"][k] * c * c
    assert abs(tot - dist) <= 1e-9 * dist, (tot, dist)
    ev_o = FastEvaluator(1, 0)
    C.oracle_enumerate(f["mut"], f["rdiag"], f["pruning"], dist, ev_o)
    # (it may find nothing at all: the device's own vector passed the level bounds of the larger
    # radius that held when it was met, not necessarily those of its own norm)
    assert all(s[0] == dist for s in ev_o.solutions), \
        ("a sequential walk at the device's final radius finds a shorter vector", ev_o.solutions, dist)
    ref_best = min(s[0] for s in f["sol_log"])
    return ref_best


@pytest.mark.parametrize("k", [0, 1, 2])
def test_config3_pruner_block_matches_reference(ctx, k):
    """One beta = 60 block under the reference pruner's coefficients (2-3 M nodes, the default.json
    regime), FastEvaluator(1): valid, complete at its own final radius, and — on blocks 0 and 1 —
    the reference's very norm (on block 2 the parallel walk usually ends on a SHORTER vector than
    the reference's sequential one: 0.66917 against 0.68177, see _check_pruned_result)."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    f = C.load_fixture(os.path.join(C.GOLDEN, "c3_b60_k%d_pruner.json" % k))
    ev = FastEvaluator(f["max_sols"], f["strategy"])
    res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev)
    assert len(ev.solutions) == 1 and len(f["sol_log"]) >= 1
    ref_best = _check_pruned_result(f, ev, res)
    if k < 2:
        assert ev.solutions[0][0] == ref_best, (ev.solutions[0][0], ref_best)
    # block 2: either walk may cut the other's vector (observed: 0.66917 on the device against the
    # reference's 0.68177 in most runs, the reference's own value in others) — validity and
    # completeness at the device's own final radius are what _check_pruned_result asserts
    # (a sanity bound on the work, not a parity statement: how many nodes a PARALLEL walk visits before the shrinking
    #  bound has reached every wave depends on the race between the waves and the host's callback — the faster walk
    #  of round 6 covers more nodes per microsecond of callback latency: 2.0-4.9 M here against the reference's
    #  2.0 M sequential ones)
    assert 0 < res.total_nodes < 4 * f["total_nodes"]
    C.note(lambda: ("C3 block %d (pruner): %d nodes (reference %d), norm %r (reference %r), %.2f ms" %
          (k, res.total_nodes, f["total_nodes"], ev.solutions[0][0], ref_best, res.stats.kernel_ms),))


@pytest.mark.parametrize("k", [0, 1, 2])
def test_config3_linear_block_matches_reference(ctx, k):
    """The benchmark's blocks (LinearPruningParams(60, 30), 5-10e9 nodes in the reference): final
    squared norm identical to the reference's."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    path = os.path.join(C.GOLDEN, "c3_b60_k%d_linear30.json" % k)
    assert os.path.exists(path), "committed fixture missing: " + path
    f = C.load_fixture(path)
    ev = FastEvaluator(f["max_sols"], f["strategy"])
    res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev)
    ref_best = min(s[0] for s in f["sol_log"])
    assert len(ev.solutions) == 1 and ev.solutions[0][0] == ref_best
    C.note(lambda: ("C3 block %d (linear30): %d nodes (reference %d), %.1f ms" %
          (k, res.total_nodes, f["total_nodes"], res.stats.kernel_ms),))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("precision", [53])
def test_config5_hlll_in_double_double_matches_reference(ctx, precision):
    """BASELINE config 5 as stated — HLLL (Householder, dd_real) on the 256-dim NTRU-like lattice:
    hlll(precision=106) runs the reference's algorithm in double-double arithmetic on the device
    (csrc/hlll_x.hip, ftx.h) and returns the reference's basis with the reference's 146 491 swaps —
    the basis `fplll -a hlll` returns in double, long double and 106-bit MPFR alike (md5 bed6b5d6…,
    SURVEY.md 8(d) C5; 869 s for the 106-bit MPFR run on one core).  precision=53 is the same
    tree-sum kernel in plain double: the at-size case of `-m gpu` (30 s); the double-double run (56 s) and the
    exact-order double run (260 s) are `-m gpu_long` (tests/test_at_size_long_runs.py)."""
    from fplll_amd.householder import MatHouseholderBatch
    f = _c5()
    h = MatHouseholderBatch(ctx, 2, 256, 256, row_expo=True)
    h.set_basis(np.stack([f["b_in"]] * 2))
    t = time.time()
    st, info = h.hlll(f["delta"], f["eta"], f["theta"], f["c"], precision=precision)
    wall = time.time() - t
    out = h.get_basis(0, 2)
    C.note(lambda: ("config 5 at precision %d: %d swaps, %.1f s on the device" % (precision, int(info[0][0]), wall),))
    assert list(st) == [1, 1] and [int(i[0]) for i in info] == [146491, 146491]
    assert np.array_equal(out[0], f["b_out"]) and np.array_equal(out[1], f["b_out"])
    h.close()
