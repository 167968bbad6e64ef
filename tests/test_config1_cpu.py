"""BASELINE config 1 — "LLL (delta = 0.99, double) on a 60-dim knapsack lattice via fplll CPU path — GSO
correctness plumbing, no GPU" (SURVEY.md 8(d), row C1).

The lattice has 1000-bit entries (ZT = mpz): by BASELINE's own wording it never touches the device.
What it pins is the plumbing of the oracle build every other parity test leans on: `latticegen r 60
1000` of oracle/_ref reproduces the survey's input fingerprint (same GMP generator), and `fplll -a
lll` — wrapper (fplll/wrapper.cpp:281-359), and `-m fast -f double` (fplll/lll.cpp:44-164 over
MatGSO<mpz, double>) — reproduces the survey's output fingerprint."""
import hashlib
import os
import subprocess

import conftest as C

REF = os.path.join(C.ROOT, "oracle", "_ref")
IN_MD5 = "7f22a4c163a3877d3ff7baa872abc3da"    # latticegen r 60 1000
OUT_MD5 = "f4f410dc3cdb0423be923b1f789089fe"   # fplll -a lll (wrapper and fast/double)


def _run(args, stdin=None):
    out = subprocess.run(args, input=stdin, capture_output=True, timeout=300)
    assert out.returncode == 0, (args, out.stderr[-500:])
    return out.stdout


def test_config1_lll_knapsack60_md5():
    gen, cli = os.path.join(REF, "latticegen"), os.path.join(REF, "fplll")
    assert os.path.exists(gen) and os.path.exists(cli), \
        "oracle/_ref is not built (python __graft_entry__.py builds it where /root/reference exists)"
    basis = _run([gen, "r", "60", "1000"])
    assert hashlib.md5(basis).hexdigest() == IN_MD5, "GMP's generator differs from the survey's"
    for mode in ([], ["-m", "fast", "-f", "double"], ["-m", "wrapper"]):
        out = _run([cli, "-a", "lll", "-d", "0.99"] + mode, stdin=basis)
        assert hashlib.md5(out).hexdigest() == OUT_MD5, mode
