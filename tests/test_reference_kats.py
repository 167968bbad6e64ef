"""The known-answer vectors the reference's own tests hold for this path:

* tests/lattices/example_svp_in -> example_svp_out (tests/test_svp.cpp:54-103): the norm of a
  shortest vector — oracle enumeration (CPU) and the device enumerator (GPU) must find that norm;
* tests/lattices/example_dsvp_in -> example_dsvp_out (tests/test_svp.cpp:214-262): after
  svp_reduction(0, d, dual = true) the LAST dual basis vector must be as short as the given dual
  vector — the oracle's dual svp_reduction (dual enumeration + dual insertion).

Fixtures tests/golden/kat_*.json are copies of those vectors (tests/golden/make_kat_fixtures.py);
all norms are compared exactly (integers / Fractions)."""
import ctypes
import json
import os
from fractions import Fraction

import numpy as np
import pytest

import conftest as C


def load(name):
    with open(os.path.join(C.GOLDEN, "kat_%s.json" % name)) as f:
        j = json.load(f)
    return np.array(j["basis"], dtype=np.int64), [int(v) for v in j["answer"]]


def lll(basis):
    o = C.OracleGSO(basis)
    st, _ = o.lll()
    assert st == 1
    b = o.b.copy()
    o.close()
    return b


def enum_input(b):
    """mu^T / r_ii of the whole (LLL-reduced) lattice, as fplll hands them to an enumerator."""
    g = C.OracleGSO(b)
    assert g.update_all()
    d = b.shape[0]
    ex = np.asarray(g.row_expo, dtype=np.int64)
    mu, r = np.array(g.mu), np.array(g.r)
    mut = np.zeros((d, d))
    rdiag = np.zeros(d)
    for i in range(d):
        rdiag[i] = np.ldexp(r[i, i], int(2 * ex[i]))
        for j in range(i + 1, d):
            mut[i, j] = np.ldexp(mu[j, i], int(ex[j] - ex[i]))
    g.close()
    return mut, rdiag


def norm2(v):
    return sum(int(x) * int(x) for x in v)


def combine(b, coeffs):
    return [sum(int(round(c)) * int(b[i, col]) for i, c in enumerate(coeffs)) for col in range(b.shape[1])]


def test_svp_kat_oracle():
    basis, answer = load("svp")
    b = lll(basis)
    mut, rdiag = enum_input(b)
    from fplll_amd.enumeration import FastEvaluator
    ev = FastEvaluator(1, 0)
    C.oracle_enumerate(mut, rdiag, None, float(rdiag[0]) * (1 + 1e-9), ev)
    assert ev.solutions
    v = combine(b, ev.solutions[0][1])
    assert norm2(v) == norm2(answer)


@pytest.mark.gpu
def test_svp_kat_device(ctx):
    basis, answer = load("svp")
    b = lll(basis)
    mut, rdiag = enum_input(b)
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    ev = FastEvaluator(1, 0)
    enumerate_block(ctx, mut, rdiag, None, float(rdiag[0]) * (1 + 1e-9), ev)
    assert ev.solutions
    v = combine(b, ev.solutions[0][1])
    assert norm2(v) == norm2(answer)


def exact_last_gso_norm2(b):
    """||b*_{d-1}||^2 exactly (rational Gram-Schmidt)."""
    d = b.shape[0]
    rows = [[Fraction(int(x)) for x in row] for row in b]
    ortho, norms = [], []
    for i in range(d):
        v = rows[i][:]
        for j in range(i):
            m = sum(x * y for x, y in zip(rows[i], ortho[j])) / norms[j]
            v = [x - m * y for x, y in zip(v, ortho[j])]
        ortho.append(v)
        norms.append(sum(x * x for x in v))
    return norms[-1]


def exact_dual_norm2(basis, coeffs):
    """|| sum_i c_i d_i ||^2 for the dual basis d_i of `basis`: c^T G^-1 c with G the Gram matrix."""
    d = basis.shape[0]
    G = [[Fraction(sum(int(basis[i, k]) * int(basis[j, k]) for k in range(basis.shape[1]))) for j in range(d)]
         for i in range(d)]
    y = [Fraction(c) for c in coeffs]
    M = [G[i] + [y[i]] for i in range(d)]
    for c in range(d):  # Gauss-Jordan, exact
        p = next(r for r in range(c, d) if M[r][c] != 0)
        M[c], M[p] = M[p], M[c]
        inv = 1 / M[c][c]
        M[c] = [x * inv for x in M[c]]
        for r in range(d):
            if r != c and M[r][c] != 0:
                f = M[r][c]
                M[r] = [x - f * z for x, z in zip(M[r], M[c])]
    sol = [M[i][d] for i in range(d)]
    return sum(Fraction(c) * s for c, s in zip(coeffs, sol))


def test_dual_svp_kat_oracle():
    basis, answer = load("dsvp")
    target = exact_dual_norm2(basis, answer)            # the KAT: length of the given dual vector
    b = lll(basis)
    g = C.OracleGSO(b)
    lib = C.oracle_lib()
    lib.oracle_gso_svp_reduction.restype = ctypes.c_int
    lib.oracle_gso_svp_reduction.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                             ctypes.c_void_p]
    nodes = ctypes.c_uint64()
    rc = lib.oracle_gso_svp_reduction(g.h, 0, b.shape[0], 1, 0.99, 0.51, None, ctypes.byref(nodes))
    assert rc == 1 and nodes.value > 0
    out = g.b.copy()
    g.close()
    assert not np.array_equal(out, b)
    last_dual = 1 / exact_last_gso_norm2(out)            # ||d_{n-1}||^2 = 1 / ||b*_{n-1}||^2
    assert last_dual <= target
    assert last_dual == target                           # (it is exactly the KAT's length)
