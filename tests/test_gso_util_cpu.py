"""The host-side members of MatGSOInterface that BKZ callers use between reductions — get_current_slope,
get_log_det, get_root_det, get_slide_potential, adjust_radius_to_gh_bound (fplll/gso_interface.cpp:197-276)
— as the product offers them (fplll_amd/csrc/gso_util_host.hip, C ABI fphip_gso_util_*; fplll_amd.gso):
bit for bit the REAL reference's numbers on the stored r diagonal / row exponents of MatGSO<long, double>
(GSO_ROW_EXPO) of BASELINE config 3's basis.  `ref_driver gsoutil` wrote tests/golden/gsoutil_q180.json;
where oracle/_ref is built the reference is also run live."""
import json
import os
import subprocess

import numpy as np
import pytest

import conftest as C

DRV = os.path.join(C.ROOT, "oracle", "_ref", "ref_driver")
BASIS = os.path.join(C.GOLDEN, "basis_q180_seed0_lll_bkz20.txt")


def _check(j):
    from fplll_amd import gso as G
    r = np.array([float.fromhex(x) for x in j["r_diag"]])
    e = np.array(j["row_expo"], dtype=np.int64)
    d = j["d"]
    for q in j["queries"]:
        a, b, bs = q["start"], q["end"], q["block_size"]
        ca, cb = max(0, a), min(d, b)
        if cb - ca >= 2:
            assert G.current_slope(r, e, ca, cb) == float.fromhex(q["slope"]), q
        assert G.log_det(r, e, a, b) == float.fromhex(q["log_det"]), q
        rd = G.root_det(r, e, a, b)
        assert rd == float.fromhex(q["root_det"]), q
        assert G.slide_potential(r, e, ca, cb, bs) == float.fromhex(q["slide_potential"]), q
        md = float.fromhex(q["max_dist"])
        assert G.adjust_radius_to_gh_bound(md, q["expo"], cb - ca, rd, 1.1) == float.fromhex(q["adjusted_1.1"]), q
        assert G.adjust_radius_to_gh_bound(md * 1e10, q["expo"], cb - ca, rd, 1.05) == \
            float.fromhex(q["adjusted_big_1.05"]), q


def _check_predicate(j):
    from fplll_amd import gso as G
    d = j["d"]
    mu = np.array([float.fromhex(x) for x in j["mu"]]).reshape(d, d)
    r = np.array([float.fromhex(x) for x in j["r"]]).reshape(d, d)
    e = np.array(j["row_expo"], dtype=np.int64)
    assert G.is_lll_reduced(mu, r, e) == bool(j["is_lll_reduced"])
    assert G.is_lll_reduced(mu, r, e, 0.999, 0.501) == bool(j["is_lll_reduced_d0999_e0501"])


SMALL = ["q40_lll", "q40_lll_rows_reversed", "q40_bkz10"]


def test_gso_utilities_match_reference_fixture():
    with open(os.path.join(C.GOLDEN, "gsoutil_q180.json")) as f:
        _check(json.load(f))


@pytest.mark.parametrize("name", SMALL)
def test_is_lll_reduced_and_utilities_match_reference_on_small_bases(name):
    """40-dimensional bases (the stored mu / r matrices are part of the fixture): an LLL-reduced one, the same
    rows in reverse order (not reduced) and a BKZ-10 reduced one; the predicate also at delta 0.999 / eta
    0.501, which none of them meets."""
    with open(os.path.join(C.GOLDEN, "gsoutil_%s.json" % name)) as f:
        j = json.load(f)
    _check(j)
    _check_predicate(j)


@pytest.mark.parametrize("name", SMALL)
def test_is_lll_reduced_matches_reference_live(name):
    if not os.path.exists(DRV):
        pytest.skip("oracle/_ref is not built on this machine (the committed fixtures cover the predicate)")
    r = subprocess.run([DRV, "gsoutil", os.path.join(C.GOLDEN, "basis_%s.txt" % name)], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-300:]
    j = json.loads(r.stdout)
    _check(j)
    _check_predicate(j)


def test_gso_utilities_match_reference_live():
    if not os.path.exists(DRV):
        pytest.skip("oracle/_ref is not built on this machine (the committed fixture covers these functions)")
    r = subprocess.run([DRV, "gsoutil", BASIS], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-300:]
    _check(json.loads(r.stdout))


def test_without_row_exponents_the_values_are_taken_as_they_are():
    from fplll_amd import gso as G
    r = np.array([16.0, 4.0, 1.0, 0.25])
    assert G.log_det(r, None, 0, 4) == float(np.log(16.0) + np.log(4.0) + np.log(1.0) + np.log(0.25))
    import math
    assert G.root_det(r, None, 0, 4) == math.exp(G.log_det(r, None, 0, 4) / 4.0)  # (1.9999999999999998)
    e = np.array([1, 0, -1, 2], dtype=np.int64)
    assert G.log_det(np.array([4.0, 4.0, 4.0, 1 / 64.0]), e, 0, 4) == G.log_det(r, None, 0, 4)
