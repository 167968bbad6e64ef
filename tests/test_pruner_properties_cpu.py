"""Size-independent properties of the pruner's primitives (host volume engine; the device kernel computes the same
doubles, tests/test_pruner_gpu.py): facts that hold for every block size and need no fixture.

  * V_1(y) = 1 for any bound, and V_k(1, …, 1) = 1: the even simplex cut by no bound is the whole simplex
    (fplll: Pruner::relative_volume, pruner/pruner_simplex.h:34-46);
  * V_k is non-decreasing in every bound and lies in (0, 1] on smooth profiles;
  * svp_probability(no pruning) = 1, and it falls when the coefficients fall;
  * the expected number of nodes falls when the coefficients fall, and linear pruning costs less than none;
  * prune() returns feasible coefficients (non-increasing from pr[0] = 1, within (0, 1]) whose metric is what
    prune() reports, for even and odd block sizes."""
import math

import numpy as np
import pytest

import conftest as C  # noqa: F401


def _vol(y, k):
    from fplll_amd import pruner as P
    y = np.asarray(y, dtype=np.float64)[None, :]
    return float(P.volumes(y, [0], [k])[0])


@pytest.mark.parametrize("m", [1, 2, 7, 16, 20])
def test_unpruned_simplex_has_relative_volume_one(m):
    # (in DOUBLE the alternating recurrence loses about 1.3 bits per level on the all-ones bound — 5e-11 at
    #  k = 16, 6e-7 at k = 24, 7e-3 at k = 32, nonsense beyond 36: the reference's arithmetic alike, which is why
    #  fplll offers wider float types for its pruner; the searches only meet decreasing profiles, where the terms
    #  shrink.  The property is checked where double still holds it.)
    ones = np.ones(m)
    for k in range(1, m + 1):
        assert abs(_vol(ones, k) - 1.0) < 1e-15 * 2.6 ** k, (m, k)
    rng = np.random.default_rng(m)
    y = np.sort(rng.uniform(0.05, 1.0, size=m))
    assert _vol(y, 1) == 1.0


@pytest.mark.parametrize("m", [4, 12, 24])
def test_volume_is_monotone_in_the_bounds(m):
    # a smooth profile (what the searches meet): linear from 0.3 to 1
    y = np.linspace(0.3, 1.0, m)
    base = [_vol(y, k) for k in range(1, m + 1)]
    assert all(0.0 < v <= 1.0 + 1e-12 for v in base)
    assert all(base[k] <= base[k - 1] * (1 + 1e-12) for k in range(1, m)), "more levels, smaller relative volume"
    for i in range(m - 1):
        z = y.copy()
        z[i] = min(z[i] * 1.05, z[i + 1])
        for k in range(i + 2, m + 1):
            assert _vol(z, k) >= _vol(y, k) * (1 - 1e-12), (i, k)


@pytest.mark.parametrize("n", [20, 31, 40])
def test_probability_and_cost_move_with_the_coefficients(n):
    from fplll_amd import pruner as P
    ones = np.ones(n)
    assert abs(P.svp_probability(ones) - 1.0) < 1e-15 * 2.6 ** (n // 2)  # (see the note on the all-ones bound above)
    lin = np.array([1.0] + [max(0.05, 1.0 - i / n) for i in range(1, n)])
    lin = np.minimum.accumulate(lin)
    p_lin = P.svp_probability(lin)
    assert 0.0 < p_lin < 1.0
    tighter = np.minimum(lin, np.maximum(0.04, lin * 0.9))
    tighter[0] = 1.0
    assert P.svp_probability(tighter) < p_lin
    # a GSA-like profile and the Gaussian-heuristic radius
    r = np.array([math.exp(-0.08 * i) for i in range(n)])
    radius = 1.05 * math.exp(sum(math.log(x) for x in r) / n) * (math.gamma(n / 2 + 1) ** (2.0 / n)) / math.pi
    c_ones = P.enum_cost(radius, r, ones)[0]
    c_lin = P.enum_cost(radius, r, lin)[0]
    c_tight = P.enum_cost(radius, r, tighter)[0]
    assert c_ones > c_lin > c_tight > 0.0


@pytest.mark.parametrize("n,target", [(24, 0.5), (33, 0.3), (40, 0.7)])
def test_prune_returns_feasible_coefficients_and_its_own_metric(n, target):
    from fplll_amd import pruner as P
    r = np.array([math.exp(-0.07 * i) for i in range(n)])
    radius = 1.1 * math.exp(sum(math.log(x) for x in r) / n) * (math.gamma(n / 2 + 1) ** (2.0 / n)) / math.pi
    pp = P.prune(radius, 1e6, r, target, P.PRUNER_METRIC_PROBABILITY_OF_SHORTEST, P.PRUNER_GRADIENT)
    c = pp.coefficients
    assert c[0] == 1.0 and np.all(c > 0.0) and np.all(c <= 1.0)
    assert np.all(np.diff(c) <= 1e-12), "coefficients do not increase with the level"
    assert pp.expectation == P.svp_probability(c)
    assert 0.0 < pp.expectation <= 1.0
    cost, metric, levels = P.enum_cost(radius, r, c)
    assert metric == pp.expectation and abs(levels.sum() - cost) < 1e-9 * cost
    assert cost < P.enum_cost(radius, r, np.ones(n))[0]


_SCALAR_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from fplll_amd import pruner as P
z = np.load(sys.argv[2])
np.save(sys.argv[3], P.volumes(z["bounds"], z["vec"], z["k"]))
"""


@pytest.mark.parametrize("m,nvec,tiny", [(5, 3, 0), (18, 7, 0), (30, 40, 0), (64, 9, 0), (30, 11, 12), (120, 5, 0)])
def test_jobs_side_by_side_are_the_one_chain_doubles(m, nvec, tiny, tmp_path):
    """The host engine takes the jobs of a batch 8 or 16 at a time as vector lanes and divides by the FMA
    correction where the CPU has FMA (pruner_volume.hip: host_volumes); FPHIP_PRUNER_HOST_SCALAR=1 keeps the
    one-job-after-the-other loop with the divider.  Same doubles, bit for bit, for ragged batches: every depth 1..m
    of several bound vectors, in shuffled order, batch sizes that leave short last groups, a batch of two (below the
    width at which lanes are used at all); bounds whose first `tiny` entries are 1e-200 (constant terms far below
    2^-300: the group is repeated with the divider); m = 120 (deeper than the FMA quotient is used for)."""
    import os
    import subprocess
    import sys
    from fplll_amd import pruner as P
    rng = np.random.default_rng(1000 * m + nvec)
    steps = rng.uniform(0.0, 1.0, size=(nvec, m))
    bounds = np.sort(steps, axis=1)          # non-decreasing bounds in (0, 1], the last one 1
    bounds /= bounds[:, -1:]
    bounds[bounds < 1e-3] = 1e-3
    if tiny:
        bounds[:, :tiny] = 1e-200
    vec, k = np.meshgrid(np.arange(nvec), np.arange(1, m + 1), indexing="ij")
    vec, k = vec.ravel(), k.ravel()
    perm = rng.permutation(vec.size)[: vec.size - 3]          # (not a multiple of eight)
    vec, k = vec[perm].astype(np.int32), k[perm].astype(np.int32)
    wide = P.volumes(bounds, vec, k)
    pair = P.volumes(bounds, vec[:2], k[:2])
    src, dst = str(tmp_path / "in.npz"), str(tmp_path / "out.npy")
    np.savez(src, bounds=bounds, vec=vec, k=k)
    subprocess.run([sys.executable, "-c", _SCALAR_CHILD, C.ROOT, src, dst], check=True, timeout=300,
                   env=dict(os.environ, FPHIP_PRUNER_HOST_SCALAR="1"))
    scalar = np.load(dst)
    assert np.array_equal(wide.view(np.uint64), scalar.view(np.uint64))
    assert np.array_equal(pair.view(np.uint64), scalar[:2].view(np.uint64))
    assert np.all(np.isfinite(wide))
