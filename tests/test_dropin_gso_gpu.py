"""The GSO half of the drop-in, end to end: the reference's UNMODIFIED LLLReduction / BKZReduction
(oracle/_ref/libfplll.so, compiled from the reference's sources as they are) run against
fplll_hip::MatGSOHip (fplll_amd/csrc/dropin/matgso_hip.h) — fplll's own MatGSO class with the device
behind it.  libfplll_hip_gso.so sits ahead of libfplll.so in oracle/_ref/dropin_driver's link order,
so every LLLReduction<long,double>::lll call the reference's bkz.cpp makes (svp_preprocessing,
bkz.cpp:107-113; the prelude, :573) is the specialisation that runs the whole LLL loop on the GPU
(fphip_gso_lll) and mirrors b / bf / mu / r / row_expo back into the host members the reference's
inline accessors read.  The output must be the reference's own (golden fixtures)."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu

DRV = os.path.join(C.ROOT, "oracle", "_ref", "dropin_driver")


def _write_basis(b):
    f = tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False)
    f.write("[" + "\n".join("[" + " ".join(str(int(x)) for x in row) + "]" for row in b) + "]\n")
    f.close()
    return f.name


def _run(args):
    assert os.path.exists(DRV), "oracle/_ref/dropin_driver is not built (python __graft_entry__.py)"
    r = subprocess.run([DRV] + args, capture_output=True, text=True, timeout=1100)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout)
    j["b_out"] = np.array(j["b_out"], dtype=np.int64).reshape(j["d"], j["n"])
    return j


@pytest.mark.parametrize("name", ["lll_q40", "lll_q72"])
def test_reference_lll_object_runs_on_the_device(name):
    """LLLReduction(m, delta, eta, LLL_DEFAULT).lll() with m a MatGSOHip: one device call, the
    reference's reduced basis and swap count."""
    f = C.load_lll_fixture(os.path.join(C.GOLDEN, name + ".json"))
    path = _write_basis(f["b_in"])
    try:
        j = _run(["lll", path, "hip"])
    finally:
        os.unlink(path)
    assert j["status"] == 0 and j["device_calls"] == 1  # RED_SUCCESS
    assert j["n_swaps"] == f["n_swaps"]
    assert np.array_equal(j["b_out"], f["b_out"])


def test_config2_reference_bkz_driver_on_device_gso():
    """BASELINE config 2 through the reference's own BKZReduction::bkz() (fplll/bkz.cpp:522-672) on a
    MatGSOHip: BKZ-20 to convergence on the 120-dim q-ary lattice — every one of its ~14 000 lll()
    calls runs on the GPU; output basis, status and enumeration node count are the golden ones."""
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "c2_bkz20_q120.json.gz"))
    path = _write_basis(f["b_in"])
    try:
        j = _run(["bkz", path, "20", "hip"])
    finally:
        os.unlink(path)
    C.note(lambda: ("config 2 via the reference's BKZReduction on MatGSOHip (resident session): %.1f s, %d device calls "
          "(%.1f s in them, %.1f s of that the LLL kernel), %d session starts, %d rows sent up as the host's row "
          "operations; reference on one core %.2f s"
          % (j["seconds"], j["device_calls"], j["device_seconds"], j.get("kernel_seconds", 0.0),
             j.get("session_starts", -1), j.get("dirty_rows", -1), f["ref_seconds"]),))
    assert j["status"] == 0 and j["device_calls"] > 1000
    assert j["nodes"] == f["nodes"] == 10252068
    assert np.array_equal(j["b_out"], f["b_out"])
    # the session is what makes this a drop-in rather than a harness: one start, every later lll() resumes
    assert j["session_starts"] == 1


def test_config2_stateless_calls_give_the_same_run():
    """FPLLL_HIP_RESIDENT=0: every interposed lll() uploads the basis and starts from a fresh MatGSO (the
    round-3 behaviour, the A/B of the session): the same golden basis and node count, on a smaller instance of the
    same family so that the suite does not pay 90 s for it."""
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "bkz_q60_b16.json"))
    path = _write_basis(f["b_in"])
    try:
        a = _run_env(["bkz", path, "16", "hip"], {"FPLLL_HIP_RESIDENT": "0"})
        b = _run_env(["bkz", path, "16", "hip"], {"FPLLL_HIP_RESIDENT": "1"})
    finally:
        os.unlink(path)
    assert np.array_equal(a["b_out"], b["b_out"]) and a["nodes"] == b["nodes"] and a["status"] == b["status"] == 0
    assert a["session_starts"] == 0 and b["session_starts"] >= 1 and a["device_calls"] == b["device_calls"] > 10


def test_bkz_reduction_with_a_transformation_matrix_on_the_device():
    """bkz_reduction(b, u, ...): MatGSOHip(b, u = identity, ...) keeps u on the device as well (the LLL kernel applies
    its row operations to u's rows; the host's insertions go up as dirty rows of b and u).  The reference's unmodified
    bkz() on it, resident and stateless: the golden basis and node count, the host object's u, and u b_in = b_out in
    exact integers."""
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "bkz_q60_b16.json"))
    path = _write_basis(f["b_in"])
    try:
        a = _run_env(["bkz", path, "16", "hip"], {"DROPIN_U": "1"})
        s = _run_env(["bkz", path, "16", "hip"], {"DROPIN_U": "1", "FPLLL_HIP_RESIDENT": "0"})
        c = _run_env(["bkz", path, "16", "cpu"], {"DROPIN_U": "1"})
    finally:
        os.unlink(path)
    d = f["d"]
    assert a["status"] == s["status"] == c["status"] == 0 and a["nodes"] == s["nodes"] == c["nodes"] == f["nodes"]
    assert np.array_equal(a["b_out"], f["b_out"]) and np.array_equal(s["b_out"], f["b_out"])
    ua, us, uc = (np.array(x["u_out"], dtype=np.int64).reshape(d, d) for x in (a, s, c))
    assert np.array_equal(ua, uc) and np.array_equal(us, uc)
    assert np.array_equal(ua.astype(object).dot(f["b_in"].astype(object)), f["b_out"].astype(object))
    assert a["session_starts"] == 1 and a["device_calls"] == s["device_calls"] > 10


def _run_env(args, env):
    assert os.path.exists(DRV), "oracle/_ref/dropin_driver is not built (python __graft_entry__.py)"
    r = subprocess.run([DRV] + args, capture_output=True, text=True, timeout=1100, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout)
    j["b_out"] = np.array(j["b_out"], dtype=np.int64).reshape(j["d"], j["n"])
    return j


@pytest.mark.parametrize("name", ["gso_q48_p3", "gso_q64_p5"])
def test_reference_size_reduction_reaches_the_device_row_by_row(name):
    """LLLReduction::size_reduction(0, d) (lll.h:107-122 — inline, so it is the CALLER's code) calls
    babai() row by row through the PLT; with FPLLL_HIP_BABAI=1 the interposed babai
    (csrc/dropin/matgso_hip.cpp) runs each row's size reduction on the device.  Output basis = the
    reference's (the fixture's b after size_reduction), d - 1 device calls."""
    f = C.load_gso_fixture(os.path.join(C.GOLDEN, name + ".json"))
    path = _write_basis(f["b_in"])
    try:
        j = _run_env(["sizered", path, "hip"], {"FPLLL_HIP_BABAI": "1"})
        jc = _run_env(["sizered", path, "cpu"], {})
    finally:
        os.unlink(path)
    assert j["status"] == 0 and jc["status"] == 0
    assert j["device_calls"] == f["d"] - 1, j["device_calls"]
    assert np.array_equal(j["b_out"], jc["b_out"])
    assert np.array_equal(j["b_out"], f["b_out"]), "size_reduction through the device differs from the golden"
    # without the opt-in the same object size-reduces on the host (no device call)
    path = _write_basis(f["b_in"])
    try:
        j0 = _run_env(["sizered", path, "hip"], {})
    finally:
        os.unlink(path)
    assert j0["device_calls"] == 0 and np.array_equal(j0["b_out"], f["b_out"])


@pytest.mark.parametrize("name", ["hlll_q40", "hlll_q72", "hlll_r30", "hlll_u24"])
def test_reference_hlll_object_runs_on_the_device(name):
    """HLLLReduction(m, delta, eta, theta, c, LLL_DEFAULT).hlll() with m a MatHouseholderHip
    (csrc/dropin/mathouseholder_hip.h): the interposed hlll() runs the whole loop on the device
    (fphip_hh_hlll) and leaves the host members — b, and R / V / sigma / bf recomputed by the
    reference's own refresh_R_bf + update_R — in line.  The reference's reduced basis and status; the
    host object's diagonal of R equals the plain host run's."""
    f = C.load_hlll_fixture(os.path.join(C.GOLDEN, name + ".json"))
    path = _write_basis(f["b_in"])
    try:
        j = _run(["hlll", path, "hip"])
        jc = _run(["hlll", path, "cpu"])
    finally:
        os.unlink(path)
    assert j["status"] == jc["status"] == 0 and j["device_calls"] == 1
    assert np.array_equal(j["b_out"], f["b_out"])
    assert np.array_equal(jc["b_out"], f["b_out"])
    assert j["log_abs_det_R"] == jc["log_abs_det_R"], (j["log_abs_det_R"], jc["log_abs_det_R"])
    C.note(lambda: ("%s through HLLLReduction on MatHouseholderHip: %d swaps, %.3f s (%.3f s on the device); host object %.3f s"
          % (name, j["n_swaps"], j["seconds"], j["device_seconds"], jc["seconds"]),))


@pytest.mark.parametrize("variant", ["siegel", "earlyred"])
def test_lll_variants_through_the_interposed_lll(variant):
    """LLL_SIEGEL (lll.cpp:38-40,122,134) and LLL_EARLY_RED (lll.cpp:84-99, lll.h:125-140) run on the device
    since round 5: one device call through the interposed lll(), the reference's basis and swap count (the host
    object run with the same flags)."""
    f = C.load_lll_fixture(os.path.join(C.GOLDEN, "lll_q40.json"))
    path = _write_basis(f["b_in"])
    try:
        j = _run(["lll", path, "hip", variant])
        jc = _run(["lll", path, "cpu", variant])
    finally:
        os.unlink(path)
    assert j["status"] == jc["status"] == 0
    assert j["device_calls"] == 1
    assert np.array_equal(j["b_out"], jc["b_out"]) and j["n_swaps"] == jc["n_swaps"]
