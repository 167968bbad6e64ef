"""Pins the C restatement (oracle/enum_oracle.c) against golden vectors produced by the REAL
reference (oracle/_ref, tests/golden/make_fixtures.sh): per-level node counts, every eval_sol
call (distance and coefficients, in order) and the final bound must be bit-identical."""
import os

import numpy as np
import pytest

import conftest as C
from fplll_amd.enumeration import FastEvaluator


@pytest.mark.parametrize("path", C.enum_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_oracle_matches_reference_fixture(path):
    f = C.load_fixture(path)
    ev = FastEvaluator(f["max_sols"], f["strategy"])
    log = []
    nodes, final = C.oracle_enumerate(f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev, log)
    assert [int(v) for v in nodes] == f["nodes"]
    assert len(log) == len(f["sol_log"])
    for (d1, x1), (d2, x2) in zip(log, f["sol_log"]):
        assert d1 == d2 and x1 == x2
    assert final == f["final_maxdist"]


@pytest.mark.parametrize("path", C.dual_enum_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_oracle_dual_walk_matches_reference_fixture(path):
    """The dual enumeration pinned at the enumeration level: the reference's
    enumerate(..., dual = true) against the oracle's dualenum walk on the transformed inputs —
    per-level node counts, every eval_sol call in order (coefficients in enumeration order: the
    reference reverses the evaluator's vectors after the walk, enumerate.cpp:154-158), final bound."""
    f = C.load_fixture(path)
    ev = FastEvaluator(f["max_sols"], f["strategy"])
    log = []
    nodes, final = C.oracle_enumerate(f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev, log, dual=True)
    assert [int(v) for v in nodes] == f["nodes"]
    assert len(log) == len(f["sol_log"])
    for (d1, x1), (d2, x2) in zip(log, f["sol_log"]):
        assert d1 == d2 and x1 == x2
    assert final == f["final_maxdist"]


def test_fixture_md5():
    import hashlib
    with open(os.path.join(C.GOLDEN, "MD5SUMS")) as fh:
        for line in fh:
            md5, name = line.split()
            p = os.path.join(C.ROOT, name)
            assert hashlib.md5(open(p, "rb").read()).hexdigest() == md5, name


def test_evaluator_mirror_replays_reference_log():
    """The Python FastEvaluator mirror (evaluator.h:122-156) fed the reference's own eval_sol log
    must end at the reference's final bound, for all three strategies."""
    for path in C.enum_fixtures():
        f = C.load_fixture(path)
        ev = FastEvaluator(f["max_sols"], f["strategy"])
        m = f["maxdist"]
        for dist, x in f["sol_log"]:
            m = ev.eval_sol(x, dist, m)
        assert m == f["final_maxdist"], f["name"]


def test_oracle_empty_and_tiny():
    # d=2 block, radius below every nonzero vector: only the zero path is visited
    mut = np.zeros((2, 2))
    mut[0, 1] = 0.25
    rd = np.array([1.0, 1.0])
    ev = FastEvaluator(1, 0)
    nodes, final = C.oracle_enumerate(mut, rd, None, 0.5, ev)
    assert ev.empty() and final == 0.5
    assert int(nodes[0]) == 1  # the zero vector is counted at level 0 (enumerate_base.cpp:33)


@pytest.mark.parametrize("path", [p for p in C.enum_fixtures() if p.endswith("_subsols.json")],
                         ids=lambda p: os.path.basename(p)[:-5])
def test_oracle_subsolutions_match_reference(path):
    """findsubsols (enumerate_base.cpp:36-40, enumerate.cpp:241-249, evaluator.h:185-205): the final
    table — best sub-solution per offset, distance and coefficients — equals the reference's."""
    from fplll_amd.enumeration import FastEvaluator
    f = C.load_fixture(path)
    ev = FastEvaluator(f["max_sols"], f["strategy"])
    nodes, _ = C.oracle_enumerate(f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev, findsubsols=True)
    assert [int(v) for v in nodes] == f["nodes"]
    assert sorted(ev.sub_solutions) == sorted(f["subsols"])
    for o, (dist, x) in f["subsols"].items():
        assert ev.sub_solutions[o][0] == dist
        assert list(ev.sub_solutions[o][1]) == x
