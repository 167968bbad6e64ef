"""The pruner's DEVICE path (fplll_amd/csrc/pruner_volume.hip: one lane per (bound vector, k), the
polynomial of a lane as a column of LDS) against the host loop of the same recurrence and against the REAL
reference (tests/golden/prune_*.json, written by `ref_driver prunefix`):

  * fphip_pruner_volumes on the device == on the host, BIT FOR BIT, on random non-decreasing bound
    vectors up to m = 64 (blocks of 128) and ragged job lists (the kernel's +, *, / are IEEE and nothing
    is contracted: one fixed operation sequence per value);
  * prune() with the searches' batches scored by the kernel returns the reference's coefficients,
    expectation and per-level costs on every fixture (gradient descent, Nelder-Mead, both metrics, odd
    block sizes, several bases) — and the kernel really ran (engine statistics)."""
import glob
import json
import os

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu


def _hx(v):
    return np.array([float.fromhex(x) for x in v], dtype=np.float64)


@pytest.fixture(scope="module")
def engine():
    from fplll_amd import pruner as P
    e = P.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("m,nvec,seed", [(30, 300, 1), (64, 500, 2), (17, 200, 3), (1, 70, 4), (100, 130, 5)])
def test_device_volumes_equal_host_volumes_bit_for_bit(engine, m, nvec, seed):
    from fplll_amd import pruner as P
    rng = np.random.default_rng(seed)
    b = np.sort(rng.uniform(0.02, 1.0, size=(nvec, m)), axis=1)
    b[:, -1] = np.where(rng.uniform(size=nvec) < 0.5, 1.0, b[:, -1])
    jv, jk = [], []
    for v in range(nvec):
        ks = range(1, m + 1) if v % 3 else rng.integers(1, m + 1, size=2)
        for k in ks:
            jv.append(v)
            jk.append(int(k))
    perm = rng.permutation(len(jv))
    jv, jk = np.array(jv)[perm], np.array(jk)[perm]
    before = engine.stats()
    dev = P.volumes(b, jv, jk, engine)
    host = P.volumes(b, jv, jk, None)
    after = engine.stats()
    steps = int(np.sum(jk.astype(np.int64) * (jk + 1) // 2))
    if steps >= 16000:
        assert after[0] - before[0] == len(jv) and after[2] == before[2] + 1, "the batch did not go through the kernel"
    else:  # (a handful of operations: evaluated inline, FPHIP_PRUNER_MIN_DEVICE_STEPS)
        assert after[1] - before[1] == len(jv) and after[2] == before[2]
    bad = np.nonzero(dev.view(np.uint64) != host.view(np.uint64))[0]
    assert bad.size == 0, (int(bad[0]), dev[bad[0]].hex(), host[bad[0]].hex(), int(jk[bad[0]]))
    # V_1 = 1 whatever the bound; small dimensions give volumes in (0, 1] (for random bounds in high dimension
    # the alternating recurrence cancels catastrophically — in the reference's arithmetic alike; the pruner
    # only meets the smooth profiles of its searches)
    assert np.all(dev[jk == 1] == 1.0)
    if m <= 17:
        assert np.all(dev > 0) and np.all(dev <= 1.0 + 1e-9)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(C.GOLDEN, "prune_*.json"))),
                         ids=lambda p: os.path.basename(p)[:-5])
def test_prune_on_the_device_matches_reference_fixture(engine, path):
    from fplll_amd import pruner as P
    with open(path) as f:
        j = json.load(f)
    before = engine.stats()
    pp = P.prune(float.fromhex(j["radius"]), float.fromhex(j["preproc_cost"]), _hx(j["gso_r"]),
                 float.fromhex(j["target"]), j["metric"], j["flags"], engine=engine)
    after = engine.stats()
    want = _hx(j["coefficients"])
    bad = np.nonzero(pp.coefficients != want)[0]
    assert bad.size == 0, ("first differing coefficient", int(bad[0]), pp.coefficients[bad[0]], want[bad[0]])
    assert pp.expectation == float.fromhex(j["expectation"])
    assert pp.gh_factor == float.fromhex(j["gh_factor"])
    assert np.array_equal(pp.detailed_cost, _hx(j["detailed_cost"]))
    # the gradients / simplices / look-ahead batches go through the kernel, lone candidates are evaluated inline
    assert after[2] > before[2] and after[0] > before[0], \
        "the searches' batches must run on the device (stats: %r -> %r)" % (before, after)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(C.GOLDEN, "prunemulti_*.json"))),
                         ids=lambda p: os.path.basename(p)[:-5])
def test_prune_over_several_bases_on_the_device(engine, path):
    from fplll_amd import pruner as P
    with open(path) as f:
        j = json.load(f)
    rs = np.stack([_hx(j["gso_r_%d" % c]) for c in range(j["count"])])
    pp = P.prune(float.fromhex(j["radius"]), float.fromhex(j["preproc_cost"]), rs, float.fromhex(j["target"]),
                 j["metric"], j["flags"], engine=engine)
    assert np.array_equal(pp.coefficients, _hx(j["coefficients"]))
    assert pp.expectation == float.fromhex(j["expectation"])


def test_device_prune_timing_note(engine):
    """How long one prune() of a 60-dimensional block takes with the host loop and with the kernel (a note,
    not an assertion: the point of the device engine is the in-loop mode of the BKZ service)."""
    import time
    from fplll_amd import pruner as P
    with open(os.path.join(C.GOLDEN, "prune_q180_k60_b60_p05.json")) as f:
        j = json.load(f)
    args = (float.fromhex(j["radius"]), float.fromhex(j["preproc_cost"]), _hx(j["gso_r"]), float.fromhex(j["target"]),
            j["metric"], j["flags"])
    t0 = time.time()
    a = P.prune(*args)
    t1 = time.time()
    b = P.prune(*args, engine=engine)
    t2 = time.time()
    assert np.array_equal(a.coefficients, b.coefficients)
    C.note(lambda: ("prune() of a 60-dim block: host loop %.1f ms, device engine %.1f ms, stats %r"
                    % (1e3 * (t1 - t0), 1e3 * (t2 - t1), engine.stats()),))
