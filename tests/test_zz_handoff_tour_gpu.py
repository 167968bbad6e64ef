"""config 3's BKZ-60 tour on the device in hand-off mode (the at-size tour of `-m gpu`; the wave-only tour, bit for
bit the reference's, is `-m gpu_long`: tests/test_at_size_long_runs.py)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest as C  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
def test_config3_bkz60_tour_with_handoff_meets_the_reducedness_predicate():
    """config 3's tour on the device in hand-off mode (large blocks on the multi-wave enumerator): the
    output is judged by the reference's own predicates — LLL-reduced (is_lll_reduced at 256 bits), same
    lattice volume, first vector not longer and slope of log r_ii not worse (within 1 %) than the
    reference tour's output — in less than 300 s (the wave-only tour: 620 s)."""
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
