"""Joins the two long device runs that tests/test_a_configs_at_size_gpu.py started at the beginning of
the GPU suite, and compares them with the reference's goldens:

  config 3  one BKZ-60 tour with the pruner strategies of the 180-dim q-ary lattice on the device
            (fplll/bkz.cpp:274-399, 522-672): basis, status, 1 224 293 770 nodes = the reference's
  config 5  HLLL of the 256-dim NTRU-like lattice in double, the reference's summation order
            (fplll/hlll.cpp:26-173): basis, status, 146 491 swaps — the NQ = 4 instantiation of the
            exact HLLL kernel"""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest as C  # noqa: E402

pytestmark = pytest.mark.gpu


def test_config3_bkz60_tour_and_config5_hlll_match_reference():
    if "thread_c3" not in C.LONG_RUNS:
        # the starter test was deselected (-k, a file selection) or ran in another xdist worker:
        # run them here rather than report a green suite that never compared them
        import test_a_configs_at_size_gpu as A
        A.start_long_runs()
    for n in ("c3", "c5"):
        C.LONG_RUNS["thread_" + n].join(1100)
        assert not C.LONG_RUNS["thread_" + n].is_alive(), "the %s run did not finish" % n
        assert n + "_error" not in C.LONG_RUNS, C.LONG_RUNS.get(n + "_error")
    c3, c5 = C.LONG_RUNS["c3"], C.LONG_RUNS["c5"]
    C.note(lambda: ("config 3 tour: %.1f s on the device (reference %.1f s on one core), %d nodes; "
          "config 5 HLLL (double, exact order): %.1f s (reference %.1f s), %d swaps"
          % (c3["wall"], c3["ref_s"], c3["nodes"][0], c5["wall"], c5["ref_s"], c5["swaps"][0]),))
    assert c3["st"] == [c3["expect"][0]] * 2 and c3["nodes"] == [c3["expect"][1]] * 2
    assert c3["expect"][1] == 1224293770 and all(c3["basis_ok"])
    assert c5["st"] == [c5["expect"]] * 2 == [1, 1] and c5["swaps"] == [146491] * 2
    assert all(c5["basis_ok"])


def test_config3_bkz60_tour_with_handoff_meets_the_reducedness_predicate():
    """config 3's tour on the device in hand-off mode (large blocks on the multi-wave enumerator): the
    output is judged by the reference's own predicates — LLL-reduced (is_lll_reduced at 256 bits), same
    lattice volume, first vector not longer and slope of log r_ii not worse (within 1 %) than the
    reference tour's output — in less than 300 s (the wave-only tour: 620 s)."""
    # (run here, after the two background runs have been joined by the test above: its worker thread
    # makes HIP calls, and beside two other busy contexts of the same process they stall — 675 s instead
    # of 80-110 s when it shared the device with them)
    import test_a_configs_at_size_gpu as A
    A._run_config3_tour_handoff(C.LONG_RUNS)
    assert "c3h_error" not in C.LONG_RUNS, C.LONG_RUNS.get("c3h_error")
    h = C.LONG_RUNS["c3h"]
    s, r, i = h["stat"], h["ref_stat"], h["in_stat"]
    C.note(lambda: ("config 3 tour with hand-off: %.1f s on the device (wave-only: see the other test; reference %.1f s), "
          "%d nodes in %d enumerations (reference %d nodes); slope %.6f (reference %.6f, input %.6f), "
          "r00 %.6g (reference %.6g)" % (h["wall"], h["ref_s"], h["nodes"], h["calls"], h["ref_nodes"],
                                         s["slope"], r["slope"], i["slope"], s["r00"], r["r00"]),))
    assert h["st"] == h["expect_status"]
    assert s["is_lll_reduced"] == 1 and r["is_lll_reduced"] == 1
    assert abs(s["log_volume"] - r["log_volume"]) < 1e-6 * abs(r["log_volume"])
    assert s["slope"] >= r["slope"] * 1.01          # slopes are negative: not steeper by more than 1 %
    assert s["slope"] > i["slope"]                  # the tour improved the basis
    assert s["r00"] <= i["r00"]
    assert h["wall"] < 300
