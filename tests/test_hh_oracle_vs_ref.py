"""Pins oracle/hh_oracle.c against the REAL reference: MatHouseholder<Z_NR<long>,FP_NR<double>>
refresh_R_bf() + update_R() (tests/golden/hh_*.json from oracle/ref_driver.cpp `hhfix`): the lower
triangle of R (incl. diagonal) and the row exponents must be bit-identical."""
import os

import numpy as np
import pytest

import conftest as C


@pytest.mark.parametrize("path", C.hh_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_hh_oracle_matches_reference(path):
    f = C.load_hh_fixture(path)
    R, V, sigma, rexp = C.oracle_hh_update_all(f["b_in"], f["row_expo_on"])
    assert np.array_equal(rexp, f["row_expo"])
    assert np.array_equal(np.tril(R[:, :f["d"]]), f["R"])
    assert np.all(np.diag(R) >= 0)  # R_ii > 0 (tests/test_gso.cpp:82-152 checks the same)
