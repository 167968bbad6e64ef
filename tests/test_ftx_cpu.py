"""fplll_amd/csrc/ftx.h — the double-double and quad-double arithmetic of the extended-precision kernels (the device
stand-ins for the reference's FP_NR<dd_real> / FP_NR<qd_real>: libqd is absent, parity with it is unpinned) — compiled
FOR THE HOST (tests/native/ftx_host.cpp: the header's arithmetic is plain C++) and checked against mpmath: every
operation to a few units of 2^-104 / 2^-205 of the result (of the larger operand for the "sloppy" additions), nint
exactly.  The GPU suite repeats the double-double part on the device (tests/test_dd_gpu.py) and runs HLLL on it."""
import os
import subprocess

import numpy as np
import pytest

import conftest as C

mp = pytest.importorskip("mpmath")


@pytest.fixture(scope="module")
def ftx(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("ftx") / "ftx_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                           os.path.join(C.ROOT, "tests", "native", "ftx_host.cpp")])

    def run(op, a, b):
        inp = "\n".join("%d %s %s" % (op, " ".join(float(t).hex() for t in a[i]), " ".join(float(t).hex() for t in b[i]))
                        for i in range(len(a)))
        out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.strip().split("\n")
        return [[float.fromhex(t) for t in l.split()] for l in out]
    return run


def _rand(rng, n, comps, lo=-30, hi=30):
    out = np.zeros((n, 4))
    for i in range(n):
        v = mp.mpf(float(rng.standard_normal())) * mp.mpf(2) ** int(rng.integers(lo, hi))
        for k in range(1, comps):
            v = v * (1 + mp.mpf(float(rng.uniform(-1, 1))) * mp.mpf(2) ** (-55 * k))
        r = v
        for k in range(comps):
            out[i, k] = float(r)
            r -= mp.mpf(out[i, k])
    return out


def _val(x):
    return sum(mp.mpf(float(t)) for t in x)


@pytest.mark.parametrize("comps,base,eps_bits", [(4, 0, 205), (2, 10, 104)])
def test_extended_arithmetic_against_mpmath(ftx, comps, base, eps_bits):
    mp.mp.prec = 900
    rng = np.random.default_rng(11 + comps)
    n = 600
    a, b = _rand(rng, n, comps), _rand(rng, n, comps)
    eps = mp.mpf(2) ** -eps_bits
    ops = [(0, "add", lambda x, y: x + y, 1), (1, "sub", lambda x, y: x - y, 1), (2, "mul", lambda x, y: x * y, 4),
           (3, "div", lambda x, y: x / y, 8), (4, "sqrt", lambda x, y: mp.sqrt(abs(x)), 4)]
    if comps == 4:
        ops.append((6, "mul by a double", lambda x, y: x * y, 4))
    for op, name, fn, tol in ops:
        aa, bb = a.copy(), b.copy()
        if op == 4:
            neg = aa[:, 0] < 0
            aa[neg] = -aa[neg]
        if op == 6:
            bb[:, 1:] = 0
        out = ftx(base + op, aa, bb)
        worst = mp.mpf(0)
        for i in range(n):
            x, y = _val(aa[i]), _val(bb[i])
            want, got = fn(x, y), _val(out[i])
            scale = max(abs(x), abs(y)) if op < 2 else abs(want)
            worst = max(worst, abs(got - want) / scale)
        assert worst <= tol * eps, (comps, name, mp.nstr(worst / eps, 5))


def test_quad_double_nint_is_exact(ftx):
    mp.mp.prec = 900
    rng = np.random.default_rng(3)
    n = 600
    q = _rand(rng, n, 4, 0, 140)
    q[::5, 0] = np.round(q[::5, 0])               # an integral leading component: the next one decides
    q[1::5, 0] = np.floor(q[1::5, 0]) + 0.5       # ties of the leading component: the next one breaks them
    for i in range(n):                            # renormalise
        r = _val(q[i])
        for k in range(4):
            q[i, k] = float(r)
            r -= mp.mpf(q[i, k])
    out = ftx(5, q, q)
    for i in range(n):
        x, got = _val(q[i]), _val(out[i])
        assert got == mp.floor(got) and abs(got - x) <= mp.mpf(1) / 2, (list(q[i]), out[i])
