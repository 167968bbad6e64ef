import ctypes
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# results of the at-size runs (tests/test_a_configs_at_size_gpu.py helpers write here, their tests read)
LONG_RUNS = {}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "gpu_long: minutes-long at-size runs on a real MI355X, one lone wavefront each "
                                       "(NOT part of -m gpu: run with -m gpu_long; their last log is profiles/r06_gpu_long_runs.log)")


def pytest_collection_modifyitems(config, items):
    """`gpu_long` tests run only when the marker expression asks for them (-m gpu_long): `-m "not gpu"` (the CPU
    suite) and a bare `pytest tests` would otherwise select them."""
    if "gpu_long" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="minutes-long at-size run: select with -m gpu_long")
    for it in items:
        if "gpu_long" in it.keywords:
            it.add_marker(skip)


def note(make):
    """Timing / progress line of a GPU test.  `make` is a callable returning the print arguments: it
    is evaluated HERE, inside a try, so that a formatting problem in a perf line can never fail a
    parity test whose assertions passed (round 3 ended red on exactly that)."""
    try:
        print(*make())
    except Exception as e:  # noqa: never part of the assertion path
        print("note unavailable: %r" % (e,))


def hexvec(a):
    return np.array([float.fromhex(s) for s in a], dtype=np.float64)


def load_fixture(path):
    with open(path) as f:
        j = json.load(f)
    d = j["d"]
    j["mut"] = hexvec(j["mut"]).reshape(d, d)
    j["rdiag"] = hexvec(j["rdiag"])
    j["pruning"] = hexvec(j["pruning"])
    j["maxdist"] = float.fromhex(j["maxdist"])
    j["final_maxdist"] = float.fromhex(j["final_maxdist"])
    j["sol_log"] = [(float.fromhex(s["dist"]), [float(v) for v in s["x"]]) for s in j["sol_log"]]
    j["name"] = os.path.basename(path)[:-5]
    if "subsols" in j:  # final sub-solution table of the reference's evaluator (findsubsols)
        j["subsols"] = {s["offset"]: (float.fromhex(s["dist"]), [float(v) for v in s["x"]])
                        for s in j["subsols"]}
    return j


def enum_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "enum_*.json")))


def dual_enum_fixtures():
    """Dual enumerations of the real reference (tests/golden/make_fixtures.sh, REFDRV_DUAL=1): the
    transformed inputs of EnumerationDyn::enumerate, per-level counts, every eval_sol call."""
    return sorted(glob.glob(os.path.join(GOLDEN, "dualenum_*.json")))


# ---- the C restatement oracle (test infrastructure; built on demand with gcc) -------------------
_oracle = None
SOLCB = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p, ctypes.c_double,
                         ctypes.POINTER(ctypes.c_double))


def oracle_lib():
    global _oracle
    if _oracle is None:
        so = os.path.join(ROOT, "oracle", "liboracle.so")
        srcs = [os.path.join(ROOT, "oracle", f) for f in ("enum_oracle.c", "gso_oracle.c", "oracle.h")]
        if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port"])
        lib = ctypes.CDLL(so)
        lib.oracle_enumerate.restype = ctypes.c_int64
        lib.oracle_enumerate.argtypes = [
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double,
            ctypes.c_int, SOLCB, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_void_p
        ]
        lib.oracle_enumerate_dual_cb.restype = ctypes.c_int64
        lib.oracle_enumerate_dual_cb.argtypes = [
            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, SOLCB,
            ctypes.c_void_p, ctypes.c_void_p
        ]
        _oracle = lib
    return _oracle


_gmp_rng = None


def gmp_rng_lib():
    """oracle/gmp_rng.c: the reference's RandGen stream (same libgmp) for rerandomize_block."""
    global _gmp_rng
    if _gmp_rng is None:
        so = os.path.join(ROOT, "oracle", "liboracle_gmprng.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port"])
        _gmp_rng = ctypes.CDLL(so)
        _gmp_rng.oracle_gmp_rng_next.restype = ctypes.c_ulong
    return _gmp_rng


class NativeRnd(tuple):
    """(address of an fphip_rand_fn, its user pointer): what `rnd` may be for the product wrappers."""


def gmp_streams_native(batch, seed):
    """The same streams as GmpStreams, served by C (oracle/gmp_rng.c): returns ((address of an
    fphip_rand_fn, user pointer), draws()) — a Python callback per random number would dominate the
    run time of the rerandomisation fixtures.  Every call owns its streams (the long runs of
    test_a_configs_at_size_gpu.py rerandomise in a thread next to the rest of the suite)."""
    lib = gmp_rng_lib()
    lib.oracle_gmp_streams_create.restype = ctypes.c_void_p
    lib.oracle_gmp_streams_create.argtypes = [ctypes.c_int, ctypes.c_ulong]
    lib.oracle_gmp_streams_draws.restype = ctypes.c_ulonglong
    lib.oracle_gmp_streams_draws.argtypes = [ctypes.c_void_p]
    h = lib.oracle_gmp_streams_create(batch, seed)
    fn = ctypes.cast(lib.oracle_gmp_streams_next, ctypes.c_void_p)
    return NativeRnd((fn, ctypes.c_void_p(h))), (lambda: lib.oracle_gmp_streams_draws(h))


class GmpStreams:
    """One GMP generator per lattice, each RandGen::init_with_seed(seed) (nr/nr_rand.inl:20-24):
    gmp_randinit_default + gmp_randseed_ui; next(lattice, n) = gmp_urandomm_ui.  Straight from the
    libgmp the reference build links — the product takes the generator from its caller."""

    def __init__(self, batch, seed):
        import ctypes.util
        path = "/opt/conda/lib/libgmp.so"
        if not os.path.exists(path):
            path = ctypes.util.find_library("gmp")
        self.g = ctypes.CDLL(path)
        # (getattr: a name starting with two underscores would be mangled inside a class body)
        self._init = getattr(self.g, "__gmp_randinit_default")
        self._seed = getattr(self.g, "__gmp_randseed_ui")
        self._next = getattr(self.g, "__gmp_urandomm_ui")
        self._next.restype = ctypes.c_ulong
        self.states = []
        for _ in range(batch):
            buf = ctypes.create_string_buffer(64)  # gmp_randstate_t
            self._init(buf)
            self._seed(buf, ctypes.c_ulong(seed))
            self.states.append(buf)
        self.draws = 0

    def __call__(self, lattice, n):
        self.draws += 1
        return self._next(self.states[lattice], ctypes.c_ulong(n))


SUBSOLCB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                            ctypes.c_int)


def oracle_enumerate(mut, rdiag, pruning, maxdist, evaluator, log=None, findsubsols=False, dual=False):
    """Run the C oracle with a Python evaluator (same protocol as the device path).  dual: the
    dualenum recursion on already transformed inputs (enumerate.cpp:107-123)."""
    lib = oracle_lib()
    mut = np.ascontiguousarray(mut, dtype=np.float64)
    d = mut.shape[0]
    rdiag = np.ascontiguousarray(rdiag, dtype=np.float64)
    pr = None
    if pruning is not None and len(pruning):
        pruning = np.ascontiguousarray(pruning, dtype=np.float64)
        pr = pruning.ctypes.data_as(ctypes.c_void_p)
    state = {"m": float(maxdist)}

    def cb(_u, dist, sol):
        x = [sol[i] for i in range(d)]
        if log is not None:
            log.append((dist, x))
        state["m"] = float(evaluator.eval_sol(x, dist, state["m"]))
        return state["m"]

    def subcb(_u, dist, sub, offset):
        evaluator.eval_sub_sol(offset, [0.0] * offset + [sub[i] for i in range(offset, d)], dist)

    c = SOLCB(cb)
    sc = SUBSOLCB(subcb) if findsubsols else None
    nodes = np.zeros(d + 1, dtype=np.uint64)
    if dual:
        assert not findsubsols
        lib.oracle_enumerate_dual_cb(d, mut.ctypes.data_as(ctypes.c_void_p),
                                     rdiag.ctypes.data_as(ctypes.c_void_p), pr, ctypes.c_double(maxdist),
                                     c, None, nodes.ctypes.data_as(ctypes.c_void_p))
        return nodes, state["m"]
    lib.oracle_enumerate(d, mut.ctypes.data_as(ctypes.c_void_p), rdiag.ctypes.data_as(ctypes.c_void_p),
                         pr, ctypes.c_double(maxdist), 1 if findsubsols else 0, c, sc, None,
                         nodes.ctypes.data_as(ctypes.c_void_p), None, None)
    return nodes, state["m"]


def synthetic_block(d, seed, slope=0.045, radius_factor=1.05):
    """A seeded GSO-like block: mu uniform in [-1/2,1/2], log r_ii decreasing linearly (what a
    BKZ-reduced block looks like); radius^2 = radius_factor * GH^2 (Gaussian heuristic of the
    block), so tree sizes behave like real SVP instances."""
    import math
    rng = np.random.default_rng(seed)
    mu = np.tril(rng.uniform(-0.5, 0.5, size=(d, d)), -1)
    mut = np.ascontiguousarray(mu.T)  # mut[i][j] = mu(j,i), j>i
    rdiag = np.exp(-2.0 * slope * np.arange(d)) * rng.uniform(0.9, 1.1, size=d)
    rdiag = rdiag / rdiag.max()
    log_gh2 = (2.0 / d) * math.lgamma(d / 2.0 + 1.0) - math.log(math.pi) + np.log(rdiag).mean()
    maxdist = float(radius_factor * math.exp(log_gh2))
    return mut, rdiag, maxdist


def wide_block(d, seed, c, d0=70, rf=0.46):
    """A block of d > 128 rows whose tree the C oracle can walk in seconds: the 70-row block
    synthetic_block(70, seed, 0.03, rf) with d - 70 rows of r_kk = c * radius^2 (times a seeded factor in
    [0.9, 1.1]) above it and seeded mu everywhere — the levels above 70 branch a little (a coefficient +-1 or two
    survive the level bounds), every such node then descends into the 70-row tree with what is left of the radius.
    Nodes at EVERY level: above 128 (four registers per lane, stack in global memory), 64..127, below 64."""
    mut0, r0, maxdist = synthetic_block(d0, seed, 0.03, rf)
    rng = np.random.default_rng(seed + 1000)
    mu = np.tril(rng.uniform(-0.5, 0.5, size=(d, d)), -1)
    mu[:d0, :d0] = mut0.T
    rdiag = np.concatenate([r0, c * maxdist * rng.uniform(0.9, 1.1, size=d - d0)])
    return np.ascontiguousarray(mu.T), rdiag, maxdist


def wide_block_with_candidates(d, seed, d0=40, rf=1.2, cheap=0.1, dear=0.45):
    """A block of d > 128 rows WITH vectors inside the radius, found under several level-64 ancestors: the
    d0-row block synthetic_block(d0, seed, 0.03, rf) (radius^2 = rf * GH^2: a handful of vectors), above it rows of
    r_kk = dear * radius^2 except four 'cheap' levels (70, 120, 140, d - 1: r_kk = cheap * radius^2) whose
    coefficients range over a few dozen combinations at little cost.  mu is zero between the rows >= d0 (their
    centres stay 0) and seeded from every row into the columns < d0, so every combination shifts the centres of
    the small block: a closest-vector instance with what is left of the radius.  The candidates' coefficients of
    levels >= 64 are non-zero and differ from ancestor to ancestor — what the table of level-64 ancestors
    (xhi_root) has to deliver, in the rows of index > 0."""
    mut0, r0, maxdist = synthetic_block(d0, seed, 0.03, rf)
    rng = np.random.default_rng(seed + 2000)
    mu = np.zeros((d, d))
    mu[:d0, :d0] = mut0.T
    mu[d0:, :d0] = rng.uniform(-0.5, 0.5, size=(d - d0, d0))
    rdiag = np.concatenate([r0, dear * maxdist * rng.uniform(0.9, 1.1, size=d - d0)])
    for k in (70, 120, 140, d - 1):
        if d0 <= k < d:
            rdiag[k] = cheap * maxdist
    return np.ascontiguousarray(mu.T), rdiag, maxdist


@pytest.fixture(scope="session")
def ctx():
    import fplll_amd
    c = fplll_amd.Context(int(os.environ.get("LOCAL_RANK", "0")))
    yield c
    c.close()


# ---- GSO oracle wrappers ---------------------------------------------------------------------
def gso_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "gso_*.json")))


def load_gso_fixture(path):
    with open(path) as f:
        j = json.load(f)
    d, n = j["d"], j["n"]
    out = {"d": d, "n": n, "name": os.path.basename(path)[:-5], "status": j["status"]}
    out["b_in"] = np.array(j["b_in"], dtype=np.int64).reshape(d, n)
    out["b_out"] = np.array(j["b_out"], dtype=np.int64).reshape(d, n)
    for k in ("mu0", "r0", "mu1", "r1"):
        out[k] = hexvec(j[k]).reshape(d, d)
    out["row_expo0"] = np.array(j["row_expo0"], dtype=np.int64)
    out["row_expo1"] = np.array(j["row_expo1"], dtype=np.int64)
    return out


class OracleGSO:
    """ctypes handle on oracle/gso_oracle.c (MatGSO<long,double> + LLLReduction::size_reduction)."""

    def __init__(self, b, row_expo=True):
        lib = oracle_lib()
        lib.oracle_gso_create.restype = ctypes.c_void_p
        lib.oracle_gso_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        lib.oracle_gso_destroy.argtypes = [ctypes.c_void_p]
        lib.oracle_gso_update_all.argtypes = [ctypes.c_void_p]
        lib.oracle_gso_size_reduction.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_double]
        for name, rt in (("oracle_gso_mu", ctypes.POINTER(ctypes.c_double)),
                         ("oracle_gso_r", ctypes.POINTER(ctypes.c_double)),
                         ("oracle_gso_bf", ctypes.POINTER(ctypes.c_double)),
                         ("oracle_gso_b", ctypes.POINTER(ctypes.c_int64)),
                         ("oracle_gso_row_expo", ctypes.POINTER(ctypes.c_int64))):
            getattr(lib, name).restype = rt
            getattr(lib, name).argtypes = [ctypes.c_void_p]
        self.lib = lib
        b = np.ascontiguousarray(b, dtype=np.int64)
        self.d, self.n = b.shape
        self.h = lib.oracle_gso_create(self.d, self.n, b.ctypes.data_as(ctypes.c_void_p),
                                       1 if row_expo else 0)

    def update_all(self):
        return self.lib.oracle_gso_update_all(self.h)

    def size_reduction(self, kmin, kend, eta=0.51):
        return self.lib.oracle_gso_size_reduction(self.h, kmin, kend, eta)

    def update_row(self, i, last=None):
        self.lib.oracle_gso_update_row.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        return self.lib.oracle_gso_update_row(self.h, i, i if last is None else last)

    def lll(self, kmin=0, kstart=0, kend=-1, delta=0.99, eta=0.51, flags=0):
        """LLLReduction::lll (oracle/gso_oracle.c); flags = fplll's LLLFlags (LLL_EARLY_RED = 2, LLL_SIEGEL = 4).
        Returns (status, info[4])."""
        self.lib.oracle_gso_lll_flags.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                  ctypes.c_int, ctypes.c_void_p]
        info = np.zeros(4, dtype=np.int32)
        for i in range(kstart):  # the caller's precondition in the reference
            self.update_row(i)
        st = self.lib.oracle_gso_lll_flags(self.h, kmin, kstart, kend, delta, eta, flags,
                                           info.ctypes.data_as(ctypes.c_void_p))
        return st, info

    def bkz(self, block_size, delta=0.99, eta=0.51, max_loops=0, auto_abort=False):
        """BKZReduction::bkz, empty strategies (oracle/gso_oracle.c).  Returns (status, info[3])."""
        self.lib.oracle_gso_bkz.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
                                            ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p]
        info = np.zeros(3, dtype=np.int32)
        st = self.lib.oracle_gso_bkz(self.h, block_size, delta, eta,
                                     (1 if max_loops > 0 else 0) | (2 if auto_abort else 0),
                                     max_loops, info.ctypes.data_as(ctypes.c_void_p))
        return st, info

    def bkz_param(self, block_size, delta, eta, flags, max_loops, gh_factor, strategies, rng_seed):
        """BKZReduction::bkz with strategies (oracle_gso_bkz_param).  strategies: the flattened dict
        of a bkzs_* fixture or None.  Returns (status, info[5])."""
        lib = self.lib

        class Strat(ctypes.Structure):
            _fields_ = [("max_block_size", ctypes.c_int), ("pre_off", ctypes.c_void_p),
                        ("pre", ctypes.c_void_p), ("prune_off", ctypes.c_void_p),
                        ("prune_gh", ctypes.c_void_p), ("prune_exp", ctypes.c_void_p),
                        ("coeff_off", ctypes.c_void_p), ("coeff", ctypes.c_void_p)]

        keep = []
        sp = None
        if strategies is not None:
            def arr(key, dt):
                a = np.ascontiguousarray(strategies[key], dtype=dt)
                if a.size == 0:
                    a = np.zeros(1, dtype=dt)
                keep.append(a)
                return a.ctypes.data
            st = Strat(int(strategies["max_block_size"]), arr("pre_off", np.int32), arr("pre", np.int32),
                       arr("prune_off", np.int32), arr("prune_gh", np.float64),
                       arr("prune_exp", np.float64), arr("coeff_off", np.int32),
                       arr("coeff", np.float64))
            keep.append(st)
            sp = ctypes.byref(st)
        rng = gmp_rng_lib()
        rng.oracle_gmp_rng_seed(ctypes.c_ulong(rng_seed))
        fn = ctypes.cast(rng.oracle_gmp_rng_next, ctypes.c_void_p)
        lib.oracle_gso_bkz_param.restype = ctypes.c_int
        lib.oracle_gso_bkz_param.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double,
                                             ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p]
        info = np.zeros(5, dtype=np.int32)
        status = lib.oracle_gso_bkz_param(self.h, block_size, delta, eta, flags, max_loops, gh_factor,
                                          sp, fn, None, info.ctypes.data_as(ctypes.c_void_p))
        return status, info

    def bkz_param_inloop(self, block_size, delta, eta, flags, max_loops, gh_factor, strategies, rng_seed, inloop):
        """oracle_gso_bkz_param with the in-loop pruning hook installed: every top-level primal block of at
        least inloop["min_block"] rows is pruned by the PRODUCT's pruner (fplll_amd.pruner.prune on the host
        volume engine) on its own r-profile — the CPU-side twin of FPHIP_BKZ_PRUNE_IN_LOOP.  Returns
        (status, info[5], number of prune() calls)."""
        from fplll_amd import pruner as P
        HOOK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))
        calls = [0]

        def hook(_user, bs, gso_r, radius, coeffs, expectation):
            calls[0] += 1
            r = np.array([gso_r[i] for i in range(bs)], dtype=np.float64)
            try:
                pp = P.prune(radius, inloop["preproc_cost"], r, inloop["target"],
                             P.PRUNER_METRIC_PROBABILITY_OF_SHORTEST, inloop["pruner_flags"])
            except Exception:
                return 0
            for i in range(bs):
                coeffs[i] = float(pp.coefficients[i])
            expectation[0] = float(pp.expectation)
            return 1
        cb = HOOK(hook)
        self.lib.oracle_gso_bkz_set_inloop.restype = None
        self.lib.oracle_gso_bkz_set_inloop.argtypes = [HOOK, ctypes.c_void_p, ctypes.c_int]
        self.lib.oracle_gso_bkz_set_inloop(cb, None, int(inloop["min_block"]))
        try:
            st, info = self.bkz_param(block_size, delta, eta, flags, max_loops, gh_factor, strategies, rng_seed)
        finally:
            self.lib.oracle_gso_bkz_set_inloop(ctypes.cast(None, HOOK), None, 0)
        return st, info, calls[0]

    def _arr(self, fn, shape, dtype):
        p = getattr(self.lib, fn)(self.h)
        return np.ctypeslib.as_array(p, shape=shape).astype(dtype).copy()

    @property
    def mu(self):
        return np.tril(self._arr("oracle_gso_mu", (self.d, self.d), np.float64), -1)

    @property
    def r(self):
        return np.tril(self._arr("oracle_gso_r", (self.d, self.d), np.float64))

    @property
    def b(self):
        return self._arr("oracle_gso_b", (self.d, self.n), np.int64)

    @property
    def bf(self):
        return self._arr("oracle_gso_bf", (self.d, self.n), np.float64)

    @property
    def row_expo(self):
        return self._arr("oracle_gso_row_expo", (self.d,), np.int64)

    def close(self):
        if self.h:
            self.lib.oracle_gso_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- LLL fixtures ---------------------------------------------------------------------------------
REF_STATUS_TO_OURS = {0: 1, 2: 0, 3: -1, 4: -3}  # RedStatus (defs.h:153-169) -> kernel / oracle code


def lll_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "lll_*.json")))


def load_lll_fixture(path):
    with open(path) as f:
        j = json.load(f)
    d, n = j["d"], j["n"]
    out = {k: j[k] for k in ("d", "n", "kmin", "kstart", "kend", "final_kappa", "n_swaps", "zeros")}
    out["name"] = os.path.basename(path)[:-5]
    out["status"] = REF_STATUS_TO_OURS[j["ref_status"]]
    out["delta"] = float.fromhex(j["delta"])
    out["eta"] = float.fromhex(j["eta"])
    out["b_in"] = np.array(j["b_in"], dtype=np.int64).reshape(d, n)
    out["b_out"] = np.array(j["b_out"], dtype=np.int64).reshape(d, n)
    out["flags"] = int(j.get("flags", 0))  # fplll's LLLFlags of the run (LLL_SIEGEL = 4)
    if "u_out" in j:  # the run kept the transformation matrix (MatGSO(b, u = identity, ...)): b_out = u_out b_in
        out["u_out"] = np.array(j["u_out"], dtype=np.int64).reshape(d, d)
    return out


# ---- BKZ fixtures ---------------------------------------------------------------------------------
BKZ_REF_STATUS_TO_OURS = {0: 1, 8: 8, 7: 7}  # RED_SUCCESS, RED_BKZ_LOOPS_LIMIT, RED_BKZ_TIME_LIMIT


def bkz_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "bkz_*.json")))


def bkz_strategy_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "bkzs_*.json")))


def load_bkz_fixture(path):
    if path.endswith(".gz"):
        import gzip
        with gzip.open(path, "rt") as f:
            j = json.load(f)
    else:
        with open(path) as f:
            j = json.load(f)
    d, n = j["d"], j["n"]
    out = {k: j[k] for k in ("d", "n", "block_size", "max_loops", "nodes")}
    if "strategies" in j:
        S = dict(j["strategies"])
        for k in ("prune_gh", "prune_exp", "coeff"):
            S[k] = [float.fromhex(v) for v in S[k]]
        out["strategies"] = S
    if "flags" in j:
        out["flags"] = j["flags"]
        out["gh_factor"] = float.fromhex(j["gh_factor"])
        out["rng_seed"] = j["rng_seed"]
    out["auto_abort"] = bool(j.get("auto_abort", 0))
    if "inloop" in j:  # in-loop pruning (ref_driver bkzfix with REFDRV_INLOOP)
        il = j["inloop"]
        out["inloop"] = dict(preproc_cost=float.fromhex(il["preproc_cost"]), target=float.fromhex(il["target"]),
                             min_block=il["min_block"], pruner_flags=il["pruner_flags"],
                             prune_calls=il["prune_calls"], prune_failures=il["prune_failures"])
    out["name"] = os.path.basename(path)[:-5]
    out["ref_seconds"] = j.get("ref_seconds")
    out["status"] = BKZ_REF_STATUS_TO_OURS[j["ref_status"]]
    out["delta"] = float.fromhex(j["delta"])
    out["eta"] = float.fromhex(j["eta"])
    out["b_in"] = np.array(j["b_in"], dtype=np.int64).reshape(d, n)
    out["b_out"] = np.array(j["b_out"], dtype=np.int64).reshape(d, n)
    return out


# ---- HLLL fixtures / oracle -----------------------------------------------------------------------
HLLL_REF_STATUS_TO_OURS = {0: 1, 11: -4, 10: -5}  # RED_HLLL_SR_FAILURE=11, RED_HLLL_NORM_FAILURE=10


def hlll_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "hlll_*.json")))


def load_hlll_fixture(path):
    if path.endswith(".gz"):
        import gzip
        with gzip.open(path, "rt") as f:
            j = json.load(f)
    else:
        with open(path) as f:
            j = json.load(f)
    d, n = j["d"], j["n"]
    out = {"d": d, "n": n, "name": os.path.basename(path)[:-5],
           "status": HLLL_REF_STATUS_TO_OURS[j["ref_status"]], "ref_seconds": j.get("ref_seconds")}
    for k in ("delta", "eta", "theta", "c"):
        out[k] = float.fromhex(j[k])
    out["b_in"] = np.array(j["b_in"], dtype=np.int64).reshape(d, n)
    out["b_out"] = np.array(j["b_out"], dtype=np.int64).reshape(d, n)
    return out


def oracle_hlll(b, delta=0.99, eta=0.51, theta=0.001, c=0.1, row_expo=True):
    """HLLLReduction::hlll (oracle/hh_oracle.c).  Returns (status, reduced basis, info[2])."""
    lib = oracle_lib()
    lib.oracle_hlll.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                ctypes.c_void_p]
    out = np.ascontiguousarray(b, dtype=np.int64).copy()
    info = np.zeros(2, dtype=np.int32)
    st = lib.oracle_hlll(out.shape[0], out.shape[1], out.ctypes.data_as(ctypes.c_void_p),
                         1 if row_expo else 0, delta, eta, theta, c,
                         info.ctypes.data_as(ctypes.c_void_p))
    return st, out, info


# ---- Householder oracle wrappers ----------------------------------------------------------------
def hh_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "hh_*.json")))


def load_hh_fixture(path):
    with open(path) as f:
        j = json.load(f)
    d, n = j["d"], j["n"]
    return {"d": d, "n": n, "name": os.path.basename(path)[:-5], "row_expo_on": j["row_expo_on"],
            "b_in": np.array(j["b_in"], dtype=np.int64).reshape(d, n),
            "R": hexvec(j["R"]).reshape(d, d), "row_expo": np.array(j["row_expo"], dtype=np.int64)}


def hhsr_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "hhsr_*.json")))


def load_hhsr_fixture(path):
    """`hhsr` fixtures (oracle/ref_driver.cpp): MatHouseholder::size_reduce(kappa, end, start) after update_R()."""
    with open(path) as f:
        j = json.load(f)
    d, n = j["d"], j["n"]
    return {"d": d, "n": n, "name": os.path.basename(path)[:-5], "row_expo_on": j["row_expo_on"],
            "kappa": j["kappa"], "end": j["end"], "start": j["start"], "reduced": j["reduced"],
            "b_in": np.array(j["b_in"], dtype=np.int64).reshape(d, n),
            "b_row": np.array(j["b_row"], dtype=np.int64), "R_row": hexvec(j["R_row"]),
            "row_expo": np.array(j["row_expo"], dtype=np.int64)}


def oracle_hh_size_reduce(b, row_expo_on, kappa, end, start):
    """oracle/hh_oracle.c: returns (flag, b after, R d×n after, row_expo)."""
    lib = oracle_lib()
    b = np.ascontiguousarray(b, dtype=np.int64).copy()
    d, n = b.shape
    R = np.zeros((d, n))
    rexp = np.zeros(d, dtype=np.int64)
    lib.oracle_hh_size_reduce.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 4 + \
        [ctypes.c_void_p] * 2
    flag = lib.oracle_hh_size_reduce(d, n, b.ctypes.data_as(ctypes.c_void_p), int(row_expo_on), int(kappa), int(end),
                                     int(start), R.ctypes.data_as(ctypes.c_void_p), rexp.ctypes.data_as(ctypes.c_void_p))
    return flag, b, R, rexp


def oracle_hh_update_all(b, row_expo_on):
    """oracle/hh_oracle.c: returns (R d×n, V d×n, sigma d, row_expo d)."""
    lib = oracle_lib()
    b = np.ascontiguousarray(b, dtype=np.int64)
    d, n = b.shape
    R = np.zeros((d, n))
    V = np.zeros((d, n))
    sigma = np.zeros(d)
    rexp = np.zeros(d, dtype=np.int64)
    lib.oracle_hh_update_all.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] + \
        [ctypes.c_int] + [ctypes.c_void_p] * 4
    lib.oracle_hh_update_all(d, n, b.ctypes.data_as(ctypes.c_void_p), int(row_expo_on),
                             R.ctypes.data_as(ctypes.c_void_p), V.ctypes.data_as(ctypes.c_void_p),
                             sigma.ctypes.data_as(ctypes.c_void_p), rexp.ctypes.data_as(ctypes.c_void_p))
    return R, V, sigma, rexp
