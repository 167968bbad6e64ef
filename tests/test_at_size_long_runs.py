"""The minutes-long at-size runs (marker `gpu_long`, NOT part of `-m gpu`): each is ONE launch that keeps one
wavefront busy for minutes, run in the foreground one after the other with a hard timeout, compared with the
reference's goldens:

  config 3  one BKZ-60 tour with the pruner strategies of the 180-dim q-ary lattice, wave-only, on the device
            (fplll/bkz.cpp:274-399, 522-672): basis, status, 1 224 293 770 nodes = the reference's (about 560 s)
  config 5  HLLL of the 256-dim NTRU-like lattice in double, the reference's summation order
            (fplll/hlll.cpp:26-173): basis, status, 146 491 swaps — the NQ = 4 instantiation of the exact HLLL
            kernel (about 260 s) — and in double-double (hlll_x.hip, 56 s)

    python -m pytest tests -m gpu_long -q        (last log: profiles/r06_gpu_long_runs.log)"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conftest as C  # noqa: E402
import test_a_configs_at_size_gpu as A  # noqa: E402

pytestmark = pytest.mark.gpu_long


@pytest.mark.timeout(1100)
def test_config3_bkz60_tour_wave_only_matches_reference():
    out = {}
    A._run_config3_tour(out)
    assert "c3_error" not in out, out.get("c3_error")
    c3 = out["c3"]
    C.note(lambda: ("config 3 tour, wave-only: %.1f s on the device (reference %.1f s on one core), %d nodes"
                    % (c3["wall"], c3["ref_s"], c3["nodes"][0]),))
    assert c3["st"] == [c3["expect"][0]] * 2 and c3["nodes"] == [c3["expect"][1]] * 2
    assert c3["expect"][1] == 1224293770 and all(c3["basis_ok"])


@pytest.mark.timeout(700)
def test_config5_hlll_exact_order_matches_reference():
    out = {}
    A._run_config5_hlll(out)
    assert "c5_error" not in out, out.get("c5_error")
    c5 = out["c5"]
    C.note(lambda: ("config 5 HLLL (double, exact order): %.1f s (reference %.1f s), %d swaps"
                    % (c5["wall"], c5["ref_s"], c5["swaps"][0]),))
    assert c5["st"] == [c5["expect"]] * 2 == [1, 1] and c5["swaps"] == [146491] * 2
    assert all(c5["basis_ok"])


@pytest.mark.timeout(400)
def test_config5_hlll_in_double_double_matches_reference():
    import fplll_amd
    ctx = fplll_amd.Context(int(os.environ.get("LOCAL_RANK", "0")))
    try:
        from fplll_amd.householder import MatHouseholderBatch
        f = A._c5()
        h = MatHouseholderBatch(ctx, 2, 256, 256, row_expo=True)
        h.set_basis(np.stack([f["b_in"]] * 2))
        st, info = h.hlll(f["delta"], f["eta"], f["theta"], f["c"], precision=106)
        out = h.get_basis(0, 2)
        C.note(lambda: ("config 5 in double-double: %d swaps, %.1f s" % (int(info[0][0]), h.last_kernel_ms / 1e3),))
        assert list(st) == [1, 1] and [int(i[0]) for i in info] == [146491, 146491]
        assert np.array_equal(out[0], f["b_out"]) and np.array_equal(out[1], f["b_out"])
        h.close()
    finally:
        ctx.close()
