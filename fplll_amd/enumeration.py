"""Host-side mirror (Python plumbing for tests and bench) of the reference's enumeration interface.

Names and semantics follow the reference so parity tests read like its own tests:
  * ``FastEvaluator(nr_solutions, strategy)``  ← fplll/enum/evaluator.h:65-205
  * ``enumerate_block(...)``                   ← what ExternalEnumeration::enumerate hands a plugin
                                                 (fplll/enum/enumerate_ext.cpp:48-148)
All arithmetic happens on the GPU behind ``fphip_enum_run`` (include/fplll_hip.h); nothing here
computes an enumeration on the CPU.
"""
import bisect
import ctypes

import numpy as np

from . import _lib

EVALSTRATEGY_BEST_N_SOLUTIONS = 0
EVALSTRATEGY_OPPORTUNISTIC_N_SOLUTIONS = 1
EVALSTRATEGY_FIRST_N_SOLUTIONS = 2


class Unsupported(RuntimeError):
    """The device layer declined the instance (fplll would fall back: enumerate_ext.cpp:88)."""


class FastEvaluator:
    """fplll::FastEvaluator<FP_NR<double>> with normExp = 0 (evaluator.h:114-205).

    ``solutions`` is kept sorted by increasing distance (the reference iterates its multimap in
    that order through begin()/end()).
    """

    def __init__(self, nr_solutions=1, strategy=EVALSTRATEGY_BEST_N_SOLUTIONS):
        if nr_solutions <= 0:
            raise ValueError("Evaluator: nr_solutions must be strictly positive!")
        if strategy not in (0, 1, 2):
            raise ValueError("Evaluator: invalid strategy")
        self.max_sols = nr_solutions
        self.strategy = strategy
        self.solutions = []  # list of (dist, tuple(coords)), ascending dist
        self.sol_count = 0
        self.sub_solutions = {}  # offset -> (dist, tuple(coords)): best per offset (findsubsols)

    def empty(self):
        return not self.solutions

    def eval_sol(self, coord, dist, max_dist):
        """process_sol (evaluator.h:122-156): returns the new max_dist."""
        self.sol_count += 1
        keys = [s[0] for s in self.solutions]
        # multimap<greater>: equal keys keep insertion order among themselves
        pos = bisect.bisect_left(keys, dist)
        self.solutions.insert(pos, (dist, tuple(coord)))
        if self.strategy == EVALSTRATEGY_BEST_N_SOLUTIONS:
            if len(self.solutions) < self.max_sols:
                return max_dist
            if len(self.solutions) > self.max_sols:
                self.solutions.pop()  # erase the longest
            return self.solutions[-1][0]
        if self.strategy == EVALSTRATEGY_OPPORTUNISTIC_N_SOLUTIONS:
            if len(self.solutions) > self.max_sols:
                self.solutions.pop()
            return dist
        # FIRST_N
        if len(self.solutions) < self.max_sols:
            return max_dist
        return 0.0


    def eval_sub_sol(self, offset, coord, dist):
        """FastEvaluator::eval_sub_sol (evaluator.h:185-205): keep the shortest per offset (a later
        candidate must be STRICTLY shorter)."""
        cur = self.sub_solutions.get(offset)
        if cur is None or dist < cur[0]:
            self.sub_solutions[offset] = (dist, tuple(coord))


class EnumResult:
    def __init__(self, nodes, stats, final_maxdist):
        self.nodes = nodes
        self.total_nodes = int(sum(int(v) for v in nodes))
        self.stats = stats
        self.final_maxdist = final_maxdist


def mut_from_mu(mu):
    """mu (lower-triangular, mu[j][i], j>i) → the transposed layout a plugin receives."""
    mu = np.asarray(mu, dtype=np.float64)
    return np.ascontiguousarray(mu.T)


def enumerate_block(ctx, mut, rdiag, pruning, maxdist, evaluator, shard_index=0, shard_count=1,
                    exchange=None, exchange_chunks=1, target_tasks=0, phase_growth=0,
                    waves_per_block=0, min_nodes_decline=0, dual=False, findsubsols=False,
                    log=None, gather=None):
    """Run one SVP enumeration on the GPU through the C ABI.

    mut[i*d+j] = mu(j,i) for j>i; rdiag, pruning (or None), maxdist normalised like the reference
    hands them to a plugin.  ``evaluator.eval_sol(coords, dist, max_dist) -> new max_dist`` is
    called (serialised) while the kernel runs.  ``exchange(local_bound, local_active) -> (bound,
    any_active)`` is the multi-GPU collective hook (RCCL all-reduce in bench.py); ``gather(block: bytes) ->
    [bytes of rank 0, bytes of rank 1, ...]`` the all-gather the work movement between ranks rides on
    (fphip_gather_cb: distributed.make_gather).
    """
    lib = ctx.lib
    mut = np.ascontiguousarray(mut, dtype=np.float64)
    d = int(round(mut.size**0.5)) if mut.ndim == 1 else mut.shape[0]
    mut = mut.reshape(d, d)
    rdiag = np.ascontiguousarray(rdiag, dtype=np.float64)
    assert rdiag.shape == (d,)
    pr_ptr = None
    if pruning is not None and len(pruning):
        pruning = np.ascontiguousarray(pruning, dtype=np.float64)
        assert pruning.shape == (d,)
        pr_ptr = pruning.ctypes.data_as(ctypes.c_void_p)
    state = {"maxdist": float(maxdist), "exc": None}

    def _cb(_user, dist, sol):
        try:
            x = [sol[i] for i in range(d)]
            if log is not None:
                log.append((dist, x))
            state["maxdist"] = float(evaluator.eval_sol(x, dist, state["maxdist"]))
        except BaseException as e:  # never let an exception cross the C boundary
            state["exc"] = e
            state["maxdist"] = 0.0
        return state["maxdist"]

    def _xc(_user, local, active, any_active):
        try:
            bound, any_ = exchange(local, bool(active))
            any_active[0] = int(bool(any_))
            return float(bound)
        except BaseException as e:
            state["exc"] = e
            any_active[0] = 0
            return 0.0

    def _gc(_user, send, send_bytes, recv, recv_cap, sizes):
        try:
            blocks = gather(ctypes.string_at(send, send_bytes) if send_bytes else b"")
            off = 0
            for r, blk in enumerate(blocks):
                if off + len(blk) > recv_cap:
                    return 2
                ctypes.memmove(recv + off, blk, len(blk))
                sizes[r] = len(blk)
                off += len(blk)
            return 0
        except BaseException as e:
            state["exc"] = e
            return 1

    def _scb(_user, dist, sub, offset):
        try:
            evaluator.eval_sub_sol(offset, [0.0] * offset + [sub[i] for i in range(offset, d)], dist)
        except BaseException as e:
            state["exc"] = e

    cb = _lib.SOL_CB(_cb)
    scb = _lib.SUBSOL_CB(_scb) if findsubsols else _lib.SUBSOL_CB()
    opts = _lib.EnumOpts()
    opts.dual = int(bool(dual))
    opts.findsubsols = int(bool(findsubsols))
    opts.shard_index = shard_index
    opts.shard_count = shard_count
    opts.exchange = _lib.EXCHANGE_CB(_xc) if exchange is not None else _lib.EXCHANGE_CB()
    opts.exchange_chunks = exchange_chunks
    opts.target_tasks = target_tasks
    opts.phase_growth = phase_growth
    opts.waves_per_block = waves_per_block
    opts.min_nodes_decline = min_nodes_decline
    opts.gather = _lib.GATHER_CB(_gc) if gather is not None else _lib.GATHER_CB()
    nodes = np.zeros(d + 1, dtype=np.uint64)
    stats = _lib.EnumStats()
    rc = lib.fphip_enum_run(ctx.handle, d, ctypes.c_double(maxdist),
                            mut.ctypes.data_as(ctypes.c_void_p),
                            rdiag.ctypes.data_as(ctypes.c_void_p), pr_ptr, ctypes.byref(opts), cb,
                            scb, None, nodes.ctypes.data_as(ctypes.c_void_p),
                            ctypes.byref(stats))
    if state["exc"] is not None:
        raise state["exc"]
    if rc == _lib.FPHIP_UNSUPPORTED:
        raise Unsupported("instance declined by the device layer (d=%d): %s" % (d, ctx.last_error()))
    if rc != _lib.FPHIP_OK:
        raise _lib.HipError("fphip_enum_run failed: %s" % ctx.last_error())
    return EnumResult(nodes, stats, state["maxdist"])
