"""ctypes mirror of the pruner's C ABI (include/fplll_hip.h: fphip_pruner_*): the reference's names
(prune, svp_probability, Pruner.single_enum_cost / measure_metric; fplll/pruner/pruner.h).  `Engine` is a
device volume engine (csrc/pruner_volume.hip): prune(..., engine=e) scores the batches of the searches
with its kernel — the same coefficients as without one.  Contains no arithmetic."""
import ctypes

import numpy as np

from . import _lib

PRUNER_METRIC_PROBABILITY_OF_SHORTEST = 0
PRUNER_METRIC_EXPECTED_SOLUTIONS = 1
PRUNER_CVP, PRUNER_START_FROM_INPUT, PRUNER_GRADIENT, PRUNER_HALF, PRUNER_SINGLE = 0x1, 0x2, 0x4, 0x20, 0x40
PRUNER_NELDER_MEAD = 0x8
PRUNER_ZEALOUS = PRUNER_GRADIENT | PRUNER_NELDER_MEAD


class PruningParams:
    """fplll's PruningParams (pruner.h:36-58): gh_factor, coefficients, expectation, metric, detailed_cost."""

    def __init__(self, gh_factor, coefficients, expectation, metric, detailed_cost):
        self.gh_factor, self.coefficients, self.expectation = gh_factor, coefficients, expectation
        self.metric, self.detailed_cost = metric, detailed_cost


def _dp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Engine:
    """fphip_pruner_engine: a stream, pinned staging and device buffers for the batched even-simplex
    volumes (one lane per (bound vector, k)).  One per host thread."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        self.h = ctypes.c_void_p()
        fn = self.lib.fphip_pruner_engine_create
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        if fn(int(device), ctypes.byref(self.h)) != _lib.FPHIP_OK:
            raise _lib.HipError("fphip_pruner_engine_create failed (no device?)")

    def stats(self):
        """(jobs evaluated by kernels, jobs evaluated inline on the host, kernel launches)"""
        a, b, c = ctypes.c_ulonglong(0), ctypes.c_ulonglong(0), ctypes.c_ulonglong(0)
        fn = self.lib.fphip_pruner_engine_stats
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_ulonglong)] * 3
        fn(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        return a.value, b.value, c.value

    def close(self):
        if getattr(self, "h", None):
            fn = self.lib.fphip_pruner_engine_destroy
            fn.restype = None
            fn.argtypes = [ctypes.c_void_p]
            fn(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def volumes(bounds, job_vec, job_k, engine=None):
    """out[j] = relative_volume(job_k[j], bounds[job_vec[j]]) (pruner_simplex.h:34-46) for a matrix of bound
    vectors; on the engine's device, or the host loop without one."""
    lib = _lib.load()
    b = np.ascontiguousarray(bounds, dtype=np.float64)
    jv = np.ascontiguousarray(job_vec, dtype=np.int32)
    jk = np.ascontiguousarray(job_k, dtype=np.int32)
    out = np.zeros(jv.size, dtype=np.float64)
    fn = lib.fphip_pruner_volumes
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_void_p]
    if fn(engine.h if engine else None, b.shape[1], b.shape[0], _dp(b), jv.size, _dp(jv), _dp(jk), _dp(out)) != _lib.FPHIP_OK:
        raise RuntimeError("fphip_pruner_volumes failed")
    return out


def prune(enumeration_radius, preproc_cost, gso_r, target=0.9, metric=PRUNER_METRIC_PROBABILITY_OF_SHORTEST,
          flags=PRUNER_GRADIENT, start=None, engine=None):
    """prune<FP_NR<double>>(pruning, radius, preproc_cost, gso_r, target, metric, flags); gso_r may be one
    profile or a list of profiles of equal length (the reference's overload for several bases)."""
    lib = _lib.load()
    r = np.ascontiguousarray(gso_r, dtype=np.float64)
    if r.ndim == 2 or engine is not None:
        return _prune_multi(lib, enumeration_radius, preproc_cost, np.atleast_2d(r), target, metric, flags, start,
                            engine)
    n = r.size
    co = np.zeros(n, dtype=np.float64)
    if start is not None:
        co[:] = np.asarray(start, dtype=np.float64)
    dc = np.zeros(n, dtype=np.float64)
    ex, gh = ctypes.c_double(0), ctypes.c_double(0)
    fn = lib.fphip_pruner_prune
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                   ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                   ctypes.c_void_p]
    rc = fn(n, _dp(r), float(enumeration_radius), float(preproc_cost), float(target), int(metric), int(flags),
            _dp(co), ctypes.byref(ex), ctypes.byref(gh), _dp(dc))
    if rc == _lib.FPHIP_UNSUPPORTED:
        raise NotImplementedError("PRUNER_VERBOSE is not offered")
    if rc != _lib.FPHIP_OK:
        raise RuntimeError("prune failed (the reference throws here: NaN / inf in a cost value, or a bad target)")
    return PruningParams(gh.value, co, ex.value, metric, dc)


def _prune_multi(lib, enumeration_radius, preproc_cost, rs, target, metric, flags, start, engine=None):
    count, n = rs.shape
    co = np.zeros(n, dtype=np.float64)
    if start is not None:
        co[:] = np.asarray(start, dtype=np.float64)
    dc = np.zeros(n, dtype=np.float64)
    ex, gh = ctypes.c_double(0), ctypes.c_double(0)
    fn = lib.fphip_pruner_prune_on
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_double,
                   ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double),
                   ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
    rc = fn(engine.h if engine else None, n, count, _dp(rs), float(enumeration_radius), float(preproc_cost),
            float(target), int(metric), int(flags), _dp(co), ctypes.byref(ex), ctypes.byref(gh), _dp(dc))
    if rc == _lib.FPHIP_UNSUPPORTED:
        raise NotImplementedError("PRUNER_VERBOSE is not offered")
    if rc != _lib.FPHIP_OK:
        raise RuntimeError("prune failed (the reference throws here: NaN / inf in a cost value, or a bad target)")
    return PruningParams(gh.value, co, ex.value, metric, dc)


def svp_probability(pr):
    """svp_probability<FP_NR<double>>(pr)."""
    lib = _lib.load()
    p = np.ascontiguousarray(pr, dtype=np.float64)
    out = ctypes.c_double(0)
    fn = lib.fphip_pruner_svp_probability
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    if fn(p.size, _dp(p), ctypes.byref(out)) != _lib.FPHIP_OK:
        raise RuntimeError("svp_probability failed")
    return out.value


def enum_cost(enumeration_radius, gso_r, pr, metric=PRUNER_METRIC_PROBABILITY_OF_SHORTEST):
    """(Pruner.single_enum_cost(pr), Pruner.measure_metric(pr), detailed_cost) for the block gso_r."""
    lib = _lib.load()
    r = np.ascontiguousarray(gso_r, dtype=np.float64)
    p = np.ascontiguousarray(pr, dtype=np.float64)
    dc = np.zeros(r.size, dtype=np.float64)
    c, m = ctypes.c_double(0), ctypes.c_double(0)
    fn = lib.fphip_pruner_enum_cost
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_int,
                   ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_void_p]
    if fn(r.size, _dp(r), float(enumeration_radius), _dp(p), int(metric), ctypes.byref(c), ctypes.byref(m),
          _dp(dc)) != _lib.FPHIP_OK:
        raise RuntimeError("enum_cost failed")
    return c.value, m.value, dc
