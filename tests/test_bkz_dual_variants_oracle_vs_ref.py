"""Pins the dual side of the BKZ oracle against the REAL reference: self-dual BKZ (BKZ_SD_VARIANT:
sd_tour = trunc_dtour + trunc_tour, bkz.cpp:401-413,443-463, with the closing hkz :627-641) and
slide reduction (BKZ_SLD_RED: slide_tour :465-520 with the slide potential
gso_interface.cpp:244-258 and the closing per-block hkz :642-663) — i.e. svp_reduction(dual = true)
(:274-358: radius 1/r of the LAST row of the block), the dual enumeration
(EnumerationDyn::enumerate's transformation enumerate.cpp:107-123,154-158 and the dualenum
recursion enumerate_base.cpp:57-61,103-105) and the dual insertions (svp_postprocessing :148-193,
_generic :240-248).  Fixtures tests/golden/bkzd_*.json (oracle/ref_driver.cpp `bkzfix` with
REFDRV_BKZ_FLAGS 0x100 / 0x200), with and without strategies.  Output basis, status and total node
count must be identical.  (oracle for the dual rows; no device path yet.)"""
import glob
import os

import numpy as np
import pytest

import conftest as C

FIXTURES = sorted(glob.glob(os.path.join(C.GOLDEN, "bkzd_*.json")))


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-5])
def test_sd_and_slide_match_reference(path):
    f = C.load_bkz_fixture(path)
    g = C.OracleGSO(f["b_in"])
    st, info = g.bkz_param(f["block_size"], f["delta"], f["eta"], f["flags"], f["max_loops"],
                           f["gh_factor"], f.get("strategies"), f["rng_seed"])
    nodes = (int(info[1]) & 0xffffffff) | ((int(info[2]) & 0xffffffff) << 32)
    assert st == f["status"]
    assert nodes == f["nodes"]
    assert np.array_equal(g.b, f["b_out"])
    assert not np.array_equal(f["b_in"], f["b_out"])
    g.close()


def test_fixture_coverage():
    flags = [C.load_bkz_fixture(p)["flags"] for p in FIXTURES]
    assert sum(1 for x in flags if x & 0x100) >= 3   # SD-BKZ
    assert sum(1 for x in flags if x & 0x200) >= 4   # slide reduction
    assert any((x & 0x200) and (x & 0x10) for x in flags)  # slide with BKZ_BOUNDED_LLL
    assert sum(1 for p in FIXTURES if "strategies" in C.load_bkz_fixture(p)) >= 2
