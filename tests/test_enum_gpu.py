"""GPU parity tests for the enumeration path: the HIP kernel, called through the C ABI
(fphip_enum_run), against (a) golden vectors from the real reference and (b) the C oracle on seeded
inputs.  Contract:
  * bound that never shrinks (BEST_N with huge N)  → per-level node counts and the multiset of
    reported solutions are IDENTICAL to the reference's;
  * shrinking bound, unpruned → identical final squared norm (the walk order of a parallel
    enumerator differs, the minimum does not);
  * shrinking bound, pruned   → identical final norm unless the reference's vector is provably cut
    by the pruning bound under the (smaller) radius the device had already reached.
"""
import json
import os
import re

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu


def _partials(mut, rdiag, x):
    """partial distances of coefficient vector x, with the reference's operation order."""
    d = len(x)
    pd = np.zeros(d + 1)
    for k in range(d - 1, -1, -1):
        c = 0.0
        for j in range(d - 1, k, -1):
            c = c - x[j] * mut[k, j]
        a = x[k] - c
        pd[k] = pd[k + 1] + a * a * rdiag[k]
    return pd


def _run(ctx, f, **kw):
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    ev = FastEvaluator(f["max_sols"], f["strategy"])
    log = []
    res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev, log=log, **kw)
    return ev, log, res


@pytest.mark.parametrize("path", C.enum_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_reference_fixture_parity(ctx, path):
    f = C.load_fixture(path)
    ev, log, res = _run(ctx, f)
    d = f["d"]
    # every reported candidate is a real vector of the stated length within the initial radius
    for dist, x in log:
        pd = _partials(f["mut"], f["rdiag"], x)
        assert pd[0] == dist
        assert 0.0 < dist <= f["maxdist"]
    fixed = f["max_sols"] >= 1000000 and f["strategy"] == 0
    if fixed:
        assert [int(v) for v in res.nodes] == f["nodes"]
        assert sorted((a, tuple(b)) for a, b in log) == sorted((a, tuple(b)) for a, b in f["sol_log"])
        return
    if f["strategy"] == 2:  # FIRST_N: any N candidates, then stop
        # a parallel walk may have a few more candidates in flight when the stop arrives
        assert len(ev.solutions) >= min(f["max_sols"], len(f["sol_log"]))
        assert (len(f["sol_log"]) > 0) == (len(log) > 0)
        if log:
            assert res.final_maxdist == 0.0
        return
    unpruned = bool(np.all(f["pruning"] == 1.0))
    if unpruned and f["strategy"] == 0:
        assert res.final_maxdist == f["final_maxdist"]
        ref_best = sorted(a for a, _ in f["sol_log"])[:f["max_sols"]]
        if ref_best:
            # the reference keeps the N shortest it saw; ours must hold the same N norms
            ref_ev = __import__("fplll_amd").FastEvaluator(f["max_sols"], 0)
            m = f["maxdist"]
            for a, b in f["sol_log"]:
                m = ref_ev.eval_sol(b, a, m)
            assert [s[0] for s in ev.solutions] == [s[0] for s in ref_ev.solutions]
        return
    if f["strategy"] == 1:  # opportunistic: order-dependent by definition; check validity only
        assert res.final_maxdist <= f["maxdist"]
        return
    # pruned + shrinking
    if res.final_maxdist != f["final_maxdist"]:
        assert f["sol_log"], "reference found nothing but we did (or vice versa)"
        ref_x = f["sol_log"][-1][1]
        pd = _partials(f["mut"], f["rdiag"], ref_x)
        cut = any(pd[k] > f["pruning"][k] * res.final_maxdist for k in range(d))
        assert cut or res.final_maxdist < f["final_maxdist"]


def _lin_pruning(d, c):
    if c is None:
        return None
    return np.maximum(0.05, 1.0 - c * np.arange(d) / d)


@pytest.mark.parametrize("d,seed,slope,rf,c", [
    (2, 1, 0.05, 3.0, None), (3, 2, 0.05, 4.0, None), (10, 3, 0.04, 2.0, None),
    (20, 4, 0.03, 1.6, None), (33, 5, 0.045, 1.3, None), (48, 6, 0.05, 1.12, 1.0),
    (60, 8, 0.05, 1.05, 1.2), (64, 7, 0.055, 1.02, 1.25)])
def test_fixed_bound_counts_equal_oracle(ctx, d, seed, slope, rf, c):
    """Sizes up to the device maximum (d = 64); bound never shrinks → exact count parity."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    mut, rdiag, maxdist = C.synthetic_block(d, seed, slope, rf)
    pruning = _lin_pruning(d, c)
    ev_o = FastEvaluator(10**9, 0)
    log_o = []
    nodes_o, _ = C.oracle_enumerate(mut, rdiag, pruning, maxdist, ev_o, log_o)
    ev = FastEvaluator(10**9, 0)
    log = []
    res = enumerate_block(ctx, mut, rdiag, pruning, maxdist, ev, log=log)
    assert [int(v) for v in res.nodes] == [int(v) for v in nodes_o]
    assert sorted((a, tuple(b)) for a, b in log) == sorted((a, tuple(b)) for a, b in log_o)


@pytest.mark.parametrize("d,seed", [(24, 11), (30, 12), (34, 13)])
def test_shrinking_unpruned_same_minimum(ctx, d, seed):
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    mut, rdiag, maxdist = C.synthetic_block(d, seed, 0.04, 1.25)
    ev_o = FastEvaluator(1, 0)
    _, final_o = C.oracle_enumerate(mut, rdiag, None, maxdist, ev_o)
    ev = FastEvaluator(1, 0)
    res = enumerate_block(ctx, mut, rdiag, None, maxdist, ev)
    assert res.final_maxdist == final_o
    assert ev.empty() == ev_o.empty()
    if not ev.empty():
        assert ev.solutions[0][0] == ev_o.solutions[0][0]


def test_pruned_fixed_bound_counts(ctx):
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    d = 40
    mut, rdiag, maxdist = C.synthetic_block(d, 21, 0.05, 1.4)
    pruning = _lin_pruning(d, 0.8)
    ev_o = FastEvaluator(10**9, 0)
    nodes_o, _ = C.oracle_enumerate(mut, rdiag, pruning, maxdist, ev_o)
    ev = FastEvaluator(10**9, 0)
    res = enumerate_block(ctx, mut, rdiag, pruning, maxdist, ev)
    assert [int(v) for v in res.nodes] == [int(v) for v in nodes_o]
    assert len(ev.solutions) == len(ev_o.solutions)


def test_many_solutions_ring_flow_control(ctx):
    """More candidates than ring slots (1024): the device must wait for the host consumer."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    d = 16
    mut, rdiag, maxdist = C.synthetic_block(d, 31, 0.0, 3.3)
    ev_o = FastEvaluator(10**9, 0)
    nodes_o, _ = C.oracle_enumerate(mut, rdiag, None, maxdist, ev_o)
    assert len(ev_o.solutions) > 3000
    ev = FastEvaluator(10**9, 0)
    res = enumerate_block(ctx, mut, rdiag, None, maxdist, ev)
    assert [int(v) for v in res.nodes] == [int(v) for v in nodes_o]
    assert [s[0] for s in ev.solutions] == [s[0] for s in ev_o.solutions]


def test_edge_cases(ctx):
    from fplll_amd.enumeration import FastEvaluator, Unsupported, enumerate_block
    # radius below every nonzero vector: only the zero path
    mut = np.zeros((2, 2))
    mut[0, 1] = 0.25
    res = enumerate_block(ctx, mut, np.array([1.0, 1.0]), None, 0.5, FastEvaluator(1, 0))
    assert [int(v) for v in res.nodes] == [1, 0, 0]
    # declined instances (fplll falls back to its own enumerator)
    for d in (1, 257, 300):  # (FPLLL_MAX_ENUM_DIM = 256 is the device's limit too since round 5)
        with pytest.raises(Unsupported):
            enumerate_block(ctx, np.zeros((d, d)), np.ones(d), None, 1.0, FastEvaluator(1, 0))
    with pytest.raises(Unsupported):  # dual + sub-solutions: the reference never asks for it
        enumerate_block(ctx, np.zeros((8, 8)), np.ones(8), None, 1.0, FastEvaluator(1, 0), dual=True,
                        findsubsols=True)
    # orthogonal basis (mu = 0): the count is the number of half-space lattice points in the ball
    d = 6
    mut, rdiag, _ = C.synthetic_block(d, 1, 0.0, 1.0)
    mut[:] = 0.0
    rdiag[:] = 1.0
    ev = FastEvaluator(10**9, 0)
    res = enumerate_block(ctx, mut, rdiag, None, 4.0, ev)
    ev_o = FastEvaluator(10**9, 0)
    nodes_o, _ = C.oracle_enumerate(mut, rdiag, None, 4.0, ev_o)
    assert [int(v) for v in res.nodes] == [int(v) for v in nodes_o]
    assert len(ev.solutions) == len(ev_o.solutions)


@pytest.mark.parametrize("d,seed,rf", [(65, 21, 0.5), (70, 22, 0.46), (100, 23, 0.22), (128, 24, 0.08),
                                       (129, 25, 0.15), (160, 26, 0.2), (200, 27, 0.3), (256, 28, 0.3)])
def test_blocks_larger_than_64_vs_oracle(ctx, d, seed, rf):
    """Two-stage walk (levels >= 64 by the top walk, then the wave-per-subtree kernel): per-level counts
    and the reported candidates are the oracle's, on seeded blocks at the chunk boundary (65), in
    between, at 128 (the last block on two registers per lane with the column stack in LDS), right above it
    (129: four registers per lane, the stack in global memory) and up to FPLLL_MAX_ENUM_DIM = 256.  The
    radii are far below the Gaussian heuristic so that the oracle finishes in seconds; candidates of
    large blocks are covered by the reference fixtures enum_d72/d80/d96."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    # (above 128 rows: conftest.wide_block, `rf` is then its c — see there)
    mut, rdiag, maxdist = C.synthetic_block(d, seed, 0.03, rf) if d <= 128 else C.wide_block(d, seed, rf)
    pruning = np.clip(np.linspace(1.0, 0.25, d)[::-1].copy(), 0.0, 1.0)[::-1].copy()
    ev = FastEvaluator(10**9, 0)
    res = enumerate_block(ctx, mut, rdiag, pruning, maxdist, ev)
    ev_o = FastEvaluator(10**9, 0)
    nodes_o, _ = C.oracle_enumerate(mut, rdiag, pruning, maxdist, ev_o)
    assert [int(v) for v in res.nodes] == [int(v) for v in nodes_o]
    assert sum(int(v) for v in nodes_o[64:]) > 0
    if d > 128:
        # the levels only the wide top walk reaches, the levels it hands down through, and the subtree kernel's
        assert sum(int(v) for v in nodes_o[128:]) > 0 and sum(int(v) for v in nodes_o[64:128]) > 1000
        assert sum(int(v) for v in nodes_o[:64]) > 10**7
    so = sorted((s[0], tuple(s[1])) for s in ev_o.solutions)
    sg = sorted((s[0], tuple(s[1])) for s in ev.solutions)
    assert sg == so
    if d > 128:
        return  # (no vector of these blocks is inside the radius: nothing shrinks)
    # shrinking radius: same final norm
    ev1, ev1o = FastEvaluator(1, 0), FastEvaluator(1, 0)
    enumerate_block(ctx, mut, rdiag, pruning, maxdist, ev1)
    C.oracle_enumerate(mut, rdiag, pruning, maxdist, ev1o)
    assert [s[0] for s in ev1.solutions] == [s[0] for s in ev1o.solutions]


@pytest.mark.parametrize("fat,d,seed,maxdist,rfat", [(2, 10, 5, 3.0, 5e-4), (1, 9, 6, 3.0, 4e-4), (3, 11, 7, 2.5, 6e-4),
                                                     (5, 12, 8, 2.2, 8e-4)])
def test_nodes_with_more_than_63_children(ctx, fat, d, seed, maxdist, rfat):
    """One level with a tiny r_kk: every node above it has 60-140 surviving children there.  The walk kernel tests
    the first 63 candidates of a node in one ballot and takes the siblings by index; beyond them (`more`), and on
    the chain of first children below the all-zero prefix (x only grows: `grow`), it tests one by one like the
    reference (enumerate_base.cpp:80-94).  Fat level 1, 2, 3: inside the walk (the breadth-first stage stops at level
    4); fat level 5: inside the breadth-first stage.  Per-level counts and the candidate multiset are the oracle's."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    mut, rdiag, _ = C.synthetic_block(d, seed, 0.0, 1.0)
    rdiag = rdiag.copy()
    rdiag[fat] = rfat
    ev, ev_o = FastEvaluator(10**9, 0), FastEvaluator(10**9, 0)
    res = enumerate_block(ctx, mut, rdiag, None, maxdist, ev)
    nodes_o, _ = C.oracle_enumerate(mut, rdiag, None, maxdist, ev_o)
    assert [int(v) for v in res.nodes] == [int(v) for v in nodes_o]
    assert int(nodes_o[fat]) > 40 * max(1, int(nodes_o[fat + 1]))  # (40-70 children per node on average, up to 140)
    assert sorted((s[0], tuple(s[1])) for s in ev.solutions) == sorted((s[0], tuple(s[1])) for s in ev_o.solutions)
    # ... and with a shrinking radius: the reference's final norm
    ev1, ev1o = FastEvaluator(1, 0), FastEvaluator(1, 0)
    enumerate_block(ctx, mut, rdiag, None, maxdist, ev1)
    C.oracle_enumerate(mut, rdiag, None, maxdist, ev1o)
    assert [s[0] for s in ev1.solutions] == [s[0] for s in ev1o.solutions]


def _dist_from(mut, rdiag, x, off=0):
    """Squared length of the projection (levels >= off) of the vector with coefficients x."""
    d = len(rdiag)
    x = np.asarray(x, dtype=np.float64)
    tot = 0.0
    for i in range(off, d):
        tot += rdiag[i] * (x[i] + float(np.dot(mut[i, i + 1:], x[i + 1:]))) ** 2
    return tot


@pytest.mark.parametrize("d,seed", [(130, 43), (160, 41), (200, 42)])
def test_wide_blocks_report_candidates_under_every_level64_ancestor(ctx, d, seed):
    """Blocks above 128 rows with vectors INSIDE the radius (conftest.wide_block_with_candidates): the candidates
    sit under several level-64 ancestors, so their coefficients of levels >= 64 come out of rows of index > 0 of
    the table the wide top walk fills (four registers per lane) and the subtree kernel reads — with ONE row
    stride, also where a block of 129..192 rows uses fewer chunks than the top walk has registers.
    Per-level counts, the candidate list (distance + all d coefficients) and the sub-solution table are the C
    oracle's; every reported vector has the reported length."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    mut, rdiag, maxdist = C.wide_block_with_candidates(d, seed)
    ev, ev_o = FastEvaluator(10**9, 0), FastEvaluator(10**9, 0)
    res = enumerate_block(ctx, mut, rdiag, None, maxdist, ev)
    nodes_o, _ = C.oracle_enumerate(mut, rdiag, None, maxdist, ev_o)
    assert [int(v) for v in res.nodes] == [int(v) for v in nodes_o]
    so = sorted((s[0], tuple(s[1])) for s in ev_o.solutions)
    sg = sorted((s[0], tuple(s[1])) for s in ev.solutions)
    assert len(so) >= 30 and len(set(s[1][64:] for s in so)) >= 5  # several ancestors, non-zero above level 64
    assert sg == so
    for dist, x in sg:
        assert abs(_dist_from(mut, rdiag, x) - dist) <= 1e-9 * dist
    # sub-solutions: the reports of the levels below 64 carry the ancestor's coefficients as well.  (The only
    # sub-solution call of the suite with >= 8192 level-64 tasks: with the split stack of the big launches its counts
    # changed from run to run — DESIGN.md section 6; sub-solution calls keep the whole stack in LDS.)
    ev, ev_o = FastEvaluator(10**9, 0), FastEvaluator(10**9, 0)
    res = enumerate_block(ctx, mut, rdiag, None, maxdist, ev, findsubsols=True)
    nodes_o, _ = C.oracle_enumerate(mut, rdiag, None, maxdist, ev_o, findsubsols=True)
    diff = [(k, int(res.nodes[k]) - int(nodes_o[k])) for k in range(d) if int(res.nodes[k]) != int(nodes_o[k])]
    assert not diff, "per-level counts differ (level, device - oracle): %s; %d candidates against %d" % (
        diff, len(ev.solutions), len(ev_o.solutions))
    assert sorted(ev.sub_solutions) == sorted(ev_o.sub_solutions)
    for o in ev_o.sub_solutions:
        dist, x = ev.sub_solutions[o]
        assert dist == ev_o.sub_solutions[o][0]
        assert abs(_dist_from(mut, rdiag, x, o) - dist) <= 1e-9 * dist, "sub-solution of level %d: wrong coefficients" % o


@pytest.mark.parametrize("path", [p for p in C.enum_fixtures() if p.endswith("_subsols.json")],
                         ids=lambda p: os.path.basename(p)[:-5])
def test_subsolutions_reference_parity(ctx, path):
    """findsubsols through the plugin protocol (extenum_cb_process_subsol): the final table of the
    evaluator — shortest sub-solution per offset — has the reference's distances; where the shortest
    is unique (no other node of that level at the same distance) also its coefficients.  The main
    results are unchanged by the option."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    f = C.load_fixture(path)
    ev = FastEvaluator(f["max_sols"], f["strategy"])
    res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev, findsubsols=True)
    assert sorted(ev.sub_solutions) == sorted(f["subsols"])
    same = 0
    for o, (dist, x) in f["subsols"].items():
        assert ev.sub_solutions[o][0] == dist
        assert all(v == 0.0 for v in ev.sub_solutions[o][1][:o])
        same += list(ev.sub_solutions[o][1]) == x
    assert same >= len(f["subsols"]) - 2  # ties at equal distance may pick another vector
    if f["max_sols"] > 10**6:
        assert [int(v) for v in res.nodes] == f["nodes"]
    assert res.final_maxdist == f["final_maxdist"] or f["strategy"] != 0


@pytest.mark.parametrize("d,seed,rf", [(40, 31, 0.95), (70, 22, 0.46)])
def test_subsolutions_vs_oracle(ctx, d, seed, rf):
    """Seeded blocks (one larger than 64: the top walk reports sub-solutions of levels >= 64)."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    mut, rdiag, maxdist = C.synthetic_block(d, seed, 0.03, rf)
    pruning = np.linspace(1.0, 0.25, d)
    ev, ev_o = FastEvaluator(10**9, 0), FastEvaluator(10**9, 0)
    res = enumerate_block(ctx, mut, rdiag, pruning, maxdist, ev, findsubsols=True)
    nodes_o, _ = C.oracle_enumerate(mut, rdiag, pruning, maxdist, ev_o, findsubsols=True)
    assert [int(v) for v in res.nodes] == [int(v) for v in nodes_o]
    assert sorted(ev.sub_solutions) == sorted(ev_o.sub_solutions)
    for o in ev_o.sub_solutions:
        assert ev.sub_solutions[o][0] == ev_o.sub_solutions[o][0]


def test_phase_parameters_do_not_change_results(ctx):
    """Cut levels / task counts / workgroup shape are scheduling only."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    f = C.load_fixture(os.path.join(C.GOLDEN, "enum_d40_lin20_fixed.json"))
    for kw in (dict(target_tasks=64, phase_growth=4), dict(target_tasks=200000, phase_growth=1000),
               dict(waves_per_block=1), dict(waves_per_block=8, target_tasks=4096)):
        ev = FastEvaluator(f["max_sols"], f["strategy"])
        res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev, **kw)
        assert [int(v) for v in res.nodes] == f["nodes"], kw


def test_sharded_counts_add_up(ctx):
    """Multi-GPU partition on one device: shards 0..3 of 4 together visit exactly the tree."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    f = C.load_fixture(os.path.join(C.GOLDEN, "enum_d48_lin30_fixed.json"))
    tot = np.zeros(f["d"] + 1, dtype=np.uint64)
    for s in range(4):
        ev = FastEvaluator(f["max_sols"], f["strategy"])
        res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev,
                              shard_index=s, shard_count=4, exchange_chunks=3)
        tot += res.nodes
    assert [int(v) for v in tot] == f["nodes"]


def test_task_buffer_overflow_is_exact(monkeypatch):
    """A tiny task buffer forces the inline-overflow path; results must not change."""
    import fplll_amd
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    monkeypatch.setenv("FPHIP_TASK_CAP", "256")
    c2 = fplll_amd.Context(0)
    try:
        f = C.load_fixture(os.path.join(C.GOLDEN, "enum_d40_lin20_fixed.json"))
        ev = FastEvaluator(f["max_sols"], f["strategy"])
        res = enumerate_block(c2, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev,
                              target_tasks=100000)
        assert [int(v) for v in res.nodes] == f["nodes"]
        assert res.stats.overflowed == 1
    finally:
        c2.close()


@pytest.mark.parametrize("d,seed,rf,cap", [(70, 22, 0.46, 2048), (65, 21, 0.5, 2048)])
def test_overflow_of_the_breadth_first_stage_above_64_rows_starts_over(monkeypatch, d, seed, rf, cap):
    """A block above 64 rows whose breadth-first stage overflows a region of the task buffer is not declined any
    more: the call starts over with the split launches, top walk included, and an overfull buffer there is walked
    inline.  Per-level counts and candidates = the oracle's; the statistics say that it happened."""
    import fplll_amd
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    monkeypatch.setenv("FPHIP_TASK_CAP", str(cap))
    monkeypatch.setenv("FPHIP_BFS_HEAVY", "1")  # (every subtree above one estimated node is expanded further)
    c2 = fplll_amd.Context(0)
    try:
        mut, rdiag, maxdist = C.synthetic_block(d, seed, 0.03, rf)
        pruning = np.clip(np.linspace(1.0, 0.25, d)[::-1].copy(), 0.0, 1.0)[::-1].copy()
        ev, ev_o = FastEvaluator(10**9, 0), FastEvaluator(10**9, 0)
        res = enumerate_block(c2, mut, rdiag, pruning, maxdist, ev, target_tasks=10**9)
        nodes_o, _ = C.oracle_enumerate(mut, rdiag, pruning, maxdist, ev_o)
        assert [int(v) for v in res.nodes] == [int(v) for v in nodes_o]
        assert sorted((s[0], tuple(s[1])) for s in ev.solutions) == sorted((s[0], tuple(s[1])) for s in ev_o.solutions)
        assert res.stats.bfs_restarts == 1, "the stage was expected to overflow with %d task slots" % cap
    finally:
        c2.close()


def test_plugin_axis_with_reference_build():
    """The reference's own test axis: install our enumerator with set_external_enumerator and
    compare against fplll's internal one in the same process (needs oracle/_ref, which travels
    with the tree; its absence is a FAILURE)."""
    import subprocess
    drv = os.path.join(C.ROOT, "oracle", "_ref", "ref_driver")
    so = os.path.join(C.ROOT, "fplll_amd", "lib", "libfplll_hip_extenum.so")
    assert os.path.exists(drv) and os.path.exists(so), \
        "oracle/_ref/ref_driver or the plugin shim is missing: run __graft_entry__.build() before " \
        "shipping the tree (the boundary rows must not go silently untested)"
    cases = [
        # n  k bits seed bkz first d pruning max_sols strategy rfac
        "80 40 12 1 0 0 32 none 1 0 0.99",
        "80 40 12 1 0 0 32 none 100000000 0 0.99",
        "100 50 14 2 20 0 40 linear:20 100000000 0 0.99",
        "80 40 12 1 0 2 36 linear:18 1 0 0.99",
    ]
    for c in cases:
        out = subprocess.run([drv, "plugin", so] + c.split(), capture_output=True, text=True,
                             timeout=600)
        assert out.returncode == 0, (c, out.stdout, out.stderr)
    # a block larger than 64 (two-stage walk), radius scaled like the enum_d80 fixture
    env = dict(os.environ, REFDRV_RADIUS_SCALE="0.45")
    for c in ("100 50 14 5 20 0 80 linear:70 100000000 0 0.99", "100 50 14 5 20 0 80 linear:70 1 0 0.99"):
        out = subprocess.run([drv, "plugin", so] + c.split(), capture_output=True, text=True,
                             timeout=600, env=env)
        assert out.returncode == 0, (c, out.stdout, out.stderr)


def test_plugin_dual_call_through_the_hook():
    """A DUAL enumeration through fplll's plugin hook (FPLLL_HIP_DUAL=1): the reference's adapter hands
    a plugin the untransformed mu / r and never reverses the solutions (enumerate_ext.cpp:57-89), so
    the shim does EnumerationDyn::enumerate's transformation itself (enumerate.cpp:107-123, 154-158).
    Driven by the reference (ref_driver plugin, REFDRV_PLUGIN_DUAL=1: svp_reduction's dual radius,
    enumerate(..., dual = true)) on a MatGSO without row exponents — the configuration in which the
    adapter's radius is right (see extenum_shim.cpp) — against fplll's internal enumerator: per-level
    node counts at fixed radius, and the solution VECTOR (orientation included) in every case."""
    import subprocess
    drv = os.path.join(C.ROOT, "oracle", "_ref", "ref_driver")
    so = os.path.join(C.ROOT, "fplll_amd", "lib", "libfplll_hip_extenum.so")
    assert os.path.exists(drv) and os.path.exists(so), "oracle/_ref/ref_driver or the plugin shim is missing"
    env = dict(os.environ, FPLLL_HIP_DUAL="1", REFDRV_PLUGIN_DUAL="1")
    for c in ("80 40 12 1 0 0 32 none 100000000 0 0.99",       # fixed radius: counts identical
              "80 40 12 1 0 3 36 linear:18 1 0 0.99",          # pruned, shrinking radius
              "100 50 14 2 20 10 40 linear:20 1 0 0.99"):
        out = subprocess.run([drv, "plugin", so] + c.split(), capture_output=True, text=True,
                             timeout=600, env=env)
        assert out.returncode == 0, (c, out.stdout, out.stderr)
        j = json.loads(out.stdout.strip().splitlines()[-1])
        assert j["ours_nodes"] > 0 and j["ours_found"] == 1, j   # it ran on the device, not declined
    # without the opt-in the shim declines a dual call and fplll's own enumerator answers
    env2 = dict(os.environ, REFDRV_PLUGIN_DUAL="1", FPLLL_HIP_STATS="1")
    out = subprocess.run([drv, "plugin", so] + "80 40 12 1 0 0 32 none 100000000 0 0.99".split(),
                         capture_output=True, text=True, timeout=600, env=env2)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert "declined" in out.stderr and " 0 enumerations on the device" in out.stderr, out.stderr


def test_plugin_in_process_multi_device():
    """FPLLL_HIP_DEVICES: the shim runs one host thread + one context per listed device inside ONE
    fplll process (host barrier + MIN at the exchange points, fphip_enum_lower_bound in between).
    "0,0" = two contexts on the one GPU of this box: the same protocol as two GPUs.  The reference
    drives it through set_external_enumerator and compares with its internal enumerator — node
    counts per level (fixed radius: identical), solutions, final radius — like the single-device
    axis above; with more than one GPU visible the same cases run on "all"."""
    import subprocess
    import fplll_amd
    drv = os.path.join(C.ROOT, "oracle", "_ref", "ref_driver")
    so = os.path.join(C.ROOT, "fplll_amd", "lib", "libfplll_hip_extenum.so")
    assert os.path.exists(drv) and os.path.exists(so), \
        "oracle/_ref/ref_driver or the plugin shim is missing: run __graft_entry__.build() before " \
        "shipping the tree (the boundary rows must not go silently untested)"
    lists = ["0,0", "0,0,0"]
    if fplll_amd.load().fphip_device_count() > 1:
        lists.append("all")
    cases = [
        ("80 40 12 1 0 0 32 none 100000000 0 0.99", None),    # fixed radius: counts identical
        ("100 50 14 2 20 0 40 linear:20 100000000 0 0.99", None),
        ("100 50 14 2 20 0 40 linear:20 1 0 0.99", None),      # shrinking radius, pruning
        ("100 50 14 5 20 0 80 linear:70 100000000 0 0.99", "0.45"),  # d > 64: top walk replicated
    ]
    for devs in lists:
        for c, scale in cases:
            env = dict(os.environ, FPLLL_HIP_DEVICES=devs)
            if scale:
                env["REFDRV_RADIUS_SCALE"] = scale
            out = subprocess.run([drv, "plugin", so] + c.split(), capture_output=True, text=True,
                                 timeout=600, env=env)
            assert out.returncode == 0, (devs, c, out.stdout[-2000:], out.stderr[-2000:])
    # the work movement between the devices of the process (the shim's all-gather among its host threads, the role
    # of make_gather between torch.distributed ranks): on by default above; here with tiny donation budgets so that
    # subtrees DO move (the totals say so), and switched off — the reference's counts either way
    c = cases[0][0]
    for mv in ("1", "0"):
        env = dict(os.environ, FPLLL_HIP_DEVICES="0,0", FPLLL_HIP_MOVE=mv, FPLLL_HIP_STATS="1", FPHIP_BUDGET="256",
                   FPHIP_MOVE_FRACTION="1000000000")
        out = subprocess.run([drv, "plugin", so] + c.split(), capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, (mv, out.stdout[-2000:], out.stderr[-2000:])
        m = re.search(r"(\d+) subtree tasks moved between devices", out.stderr)
        assert m, out.stderr[-1500:]
        assert (int(m.group(1)) > 0) == (mv == "1"), (mv, m.group(0))


def _dual_inputs(mut, rdiag):
    """EnumerationDyn::enumerate's transformation for a dual call (enumerate.cpp:107-123)."""
    d = len(rdiag)
    rd = np.zeros(d)
    mt = np.zeros((d, d))
    for i in range(d):
        rd[d - i - 1] = 1.0 / rdiag[i]
        for j in range(i + 1, d):
            mt[d - j - 1, d - i - 1] = -mut[i, j]
    return mt, rd


@pytest.mark.parametrize("d,seed,slope,rf,c", [
    (3, 21, 0.05, 4.0, None), (24, 22, 0.03, 1.5, None), (48, 23, 0.05, 1.12, 1.0),
    (64, 24, 0.055, 1.02, 1.25), (72, 25, 0.05, 1.0, 1.3), (96, 26, 0.045, 0.86, 1.45)])
def test_dual_fixed_bound_counts_equal_oracle(ctx, d, seed, slope, rf, c):
    """Dual enumeration through the C ABI (opts.dual): per-level node counts and
    the multiset of candidates equal the oracle's dualenum walk (pinned against the reference through
    the self-dual BKZ fixtures, test_bkz_dual_variants_oracle_vs_ref.py); the sizes above 64 take the
    top-walk kernel."""
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    mut, rdiag, _ = C.synthetic_block(d, seed, slope, rf)
    mt, rd = _dual_inputs(mut, rdiag)
    import math
    log_gh2 = (2.0 / d) * math.lgamma(d / 2.0 + 1.0) - math.log(math.pi) + np.log(rd).mean()
    maxdist = float(rf * math.exp(log_gh2))
    pruning = _lin_pruning(d, c)
    ev_o, log_o = FastEvaluator(10**9, 0), []
    nodes_o, _ = C.oracle_enumerate(mt, rd, pruning, maxdist, ev_o, log_o, dual=True)
    ev, log = FastEvaluator(10**9, 0), []
    res = enumerate_block(ctx, mt, rd, pruning, maxdist, ev, log=log, dual=True)
    assert [int(v) for v in res.nodes] == [int(v) for v in nodes_o]
    assert sorted((a, tuple(b)) for a, b in log) == sorted((a, tuple(b)) for a, b in log_o)
    if 24 <= d <= 64:  # ... and the dual walk is a different walk from the primal one on these inputs
        nodes_p, _ = C.oracle_enumerate(mt, rd, pruning, maxdist, FastEvaluator(10**9, 0))
        assert [int(v) for v in nodes_o] != [int(v) for v in nodes_p]


def test_dual_svp_kat_device(ctx):
    """The reference's dual-SVP known answer (tests/lattices/example_dsvp_in/out, tests/test_svp.cpp:
    214-262) through the device's dual enumeration: the shortest dual vector it finds has exactly the
    KAT's length, and the run equals the oracle's on the same transformed inputs."""
    import test_reference_kats as K
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    basis, answer = K.load("dsvp")
    target = K.exact_dual_norm2(basis, answer)
    b = K.lll(basis)
    mut, rdiag = K.enum_input(b)
    d = b.shape[0]
    mt, rd = _dual_inputs(mut, rdiag)
    maxdist = float(rd[0] * 1.0001)  # the dual vector d_{n-1} itself is inside
    ev_o = FastEvaluator(1, 0)
    _, m_o = C.oracle_enumerate(mt, rd, None, maxdist, ev_o, dual=True)
    ev = FastEvaluator(1, 0)
    res = enumerate_block(ctx, mt, rd, None, maxdist, ev, dual=True)
    assert ev.solutions and res.final_maxdist == m_o
    x = [int(v) for v in ev.solutions[0][1]][::-1]  # enumerate.cpp:154-158: reversed for dual
    assert K.exact_dual_norm2(b, x) == target


@pytest.mark.parametrize("path", C.dual_enum_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_dual_reference_fixture_parity(ctx, path):
    """Dual enumeration against the REAL reference (tests/golden/dualenum_*.json: enumerate(...,
    dual = true) of fplll, inputs transformed as EnumerationDyn::enumerate does): with a radius that
    never shrinks the per-level node counts and the multiset of candidates are the reference's
    (d = 36 and d = 72, the latter through the top-walk kernel); with BEST-1 the final norm is."""
    f = C.load_fixture(path)
    ev, log, res = _run(ctx, f, dual=True)
    for dist, x in log:
        assert 0.0 < dist <= f["maxdist"]
    if f["max_sols"] >= 1000000 and f["strategy"] == 0:
        assert [int(v) for v in res.nodes] == f["nodes"]
        assert sorted((a, tuple(b)) for a, b in log) == sorted((a, tuple(b)) for a, b in f["sol_log"])
        return
    # shrinking radius (a pruned tree is order dependent, see the module docstring): never longer
    # than the reference's result unless its vector is cut under the radius the device had reached
    if res.final_maxdist != f["final_maxdist"]:
        assert f["sol_log"] and res.final_maxdist < f["final_maxdist"]
