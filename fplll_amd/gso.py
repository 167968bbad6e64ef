"""Host-side mirror of the reference's GSO interface for a BATCH of lattices on the GPU.

``MatGSOBatch`` keeps the names of fplll's ``MatGSO`` / ``LLLReduction`` members
(update_gso, size_reduction, get_mu, get_r, row_expo — fplll/gso_interface.h, fplll/lll.h) at sweep
granularity; every number is computed by the HIP kernels behind include/fplll_hip.h.
"""
import ctypes
import os

import numpy as np

from . import _lib

LLL_DEF_ETA = 0.51
LLL_DEF_DELTA = 0.99  # fplll/defs.h:143-151


def _bind(lib):
    if getattr(lib, "_gso_bound", False):
        return
    vp = ctypes.c_void_p
    lib.fphip_gso_create.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(vp)]
    lib.fphip_gso_create.restype = ctypes.c_int
    lib.fphip_gso_destroy.argtypes = [vp]
    lib.fphip_gso_destroy.restype = None
    for name in ("fphip_gso_set_basis", "fphip_gso_get_basis"):
        getattr(lib, name).argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
        getattr(lib, name).restype = ctypes.c_int
    lib.fphip_gso_broadcast_basis.argtypes = [vp, ctypes.c_int]
    lib.fphip_gso_broadcast_basis.restype = ctypes.c_int
    lib.fphip_gso_tile_basis.argtypes = [vp, ctypes.c_int]
    lib.fphip_gso_tile_basis.restype = ctypes.c_int
    lib.fphip_gso_refresh.argtypes = [vp]
    lib.fphip_gso_refresh.restype = ctypes.c_int
    lib.fphip_gso_update.argtypes = [vp, vp]
    lib.fphip_gso_update.restype = ctypes.c_int
    lib.fphip_gso_size_reduce.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, vp]
    lib.fphip_gso_size_reduce.restype = ctypes.c_int
    lib.fphip_gso_lll.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                  ctypes.c_double, vp, vp]
    lib.fphip_gso_lll.restype = ctypes.c_int
    lib.fphip_gso_bkz.argtypes = [vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                  ctypes.c_int, vp, vp]
    lib.fphip_gso_bkz.restype = ctypes.c_int
    for name in ("fphip_gso_get_mu", "fphip_gso_get_r", "fphip_gso_get_row_expo"):
        getattr(lib, name).argtypes = [vp, ctypes.c_int, vp]
        getattr(lib, name).restype = ctypes.c_int
    lib.fphip_gso_last_kernel_ms.argtypes = [vp]
    lib.fphip_gso_last_kernel_ms.restype = ctypes.c_double
    lib._gso_bound = True


def inverse_transpose(u):
    """(u^-1)^T of a unimodular integer matrix, exactly: fraction-free Gauss-Jordan elimination (Bareiss: every
    division is exact, the diagonal ends as det(u) = +-1) on Python integers.  O(d^3) big-integer operations —
    a fraction of a second at d = 40, seconds at d = 130."""
    u = np.asarray(u)
    d = u.shape[0]
    assert u.shape == (d, d)
    A = np.empty((d, 2 * d), dtype=object)
    for i in range(d):
        for j in range(d):
            A[i, j] = int(u[i, j])
            A[i, d + j] = 1 if i == j else 0
    prev = 1
    for k in range(d):
        p = next((i for i in range(k, d) if A[i, k] != 0), None)
        if p is None:
            raise ValueError("inverse_transpose: the matrix is singular")
        if p != k:
            A[[k, p]] = A[[p, k]]
        akk = A[k, k]
        rowk = A[k].copy()
        for i in range(d):
            if i != k:
                aik = A[i, k]
                if aik == 0 and akk == prev:
                    continue  # (the row is unchanged: most rows of a transformation close to the identity)
                A[i] = (A[i] * akk - rowk * aik) // prev
        prev = akk
    det = A[d - 1, d - 1]  # (every diagonal entry is the determinant now, up to the sign of the row swaps: +-1)
    if det not in (1, -1):
        raise ValueError("inverse_transpose: not unimodular (|det| = %d)" % abs(det))
    inv = A[:, d:] * det  # adj(u) / det with det = +-1
    return inv.T.copy()


class MatGSOBatch:
    """`batch` independent d×n integer lattices, GSO state resident in HBM."""

    def __init__(self, ctx, batch, d, n, row_expo=True):
        self.ctx = ctx
        ctx.adopt(self)
        self.lib = ctx.lib
        _bind(self.lib)
        self.batch, self.d, self.n = batch, d, n
        self.h = ctypes.c_void_p()
        rc = self.lib.fphip_gso_create(ctx.handle, batch, d, n, 1 if row_expo else 0,
                                       ctypes.byref(self.h))
        if rc == _lib.FPHIP_UNSUPPORTED:
            raise NotImplementedError("d, n > 256 are not handled on the device")
        if rc != _lib.FPHIP_OK:
            raise _lib.HipError("fphip_gso_create: " + ctx.last_error())

    def _chk(self, rc, what):
        if rc != _lib.FPHIP_OK:
            raise _lib.HipError("%s: %s" % (what, self.ctx.last_error()))

    def set_basis(self, b, first=0):
        b = np.ascontiguousarray(b, dtype=np.int64)
        if b.ndim == 2:
            b = b[None]
        assert b.shape[1:] == (self.d, self.n)
        self._chk(self.lib.fphip_gso_set_basis(self.h, first, b.shape[0],
                                               b.ctypes.data_as(ctypes.c_void_p)), "set_basis")
        self._chk(self.lib.fphip_gso_refresh(self.h), "refresh")

    def broadcast_basis(self, src=0):
        self._chk(self.lib.fphip_gso_broadcast_basis(self.h, src), "broadcast_basis")
        self._chk(self.lib.fphip_gso_refresh(self.h), "refresh")

    def tile_basis(self, count):
        """lattices count.. := copies of lattices 0..count-1, cyclically (device-side copies)"""
        self._chk(self.lib.fphip_gso_tile_basis(self.h, count), "tile_basis")
        self._chk(self.lib.fphip_gso_refresh(self.h), "refresh")

    def get_basis(self, first=0, count=None):
        count = self.batch - first if count is None else count
        b = np.empty((count, self.d, self.n), dtype=np.int64)
        self._chk(self.lib.fphip_gso_get_basis(self.h, first, count,
                                               b.ctypes.data_as(ctypes.c_void_p)), "get_basis")
        return b

    def enable_transform(self, u=None):
        """MatGSO(b, u, ...) with a non-empty u (enable_transform): track the transformation matrix on the device —
        u [batch][d][d] int64, or None for the identity.  lll() then applies every row operation to u as well
        (gso.cpp:84-158); the entry points that do not raise Unsupported while it is tracked."""
        fn = self.lib.fphip_gso_enable_transform
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        if u is not None:
            u = np.ascontiguousarray(u, dtype=np.int64)
            assert u.shape == (self.batch, self.d, self.d)
        self._chk(fn(self.h, None if u is None else u.ctypes.data_as(ctypes.c_void_p)), "enable_transform")

    def get_transform(self, first=0, count=None):
        count = self.batch - first if count is None else count
        u = np.empty((count, self.d, self.d), dtype=np.int64)
        fn = self.lib.fphip_gso_get_transform
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        self._chk(fn(self.h, first, count, u.ctypes.data_as(ctypes.c_void_p)), "get_transform")
        return u

    def get_inverse_transform_t(self, first=0, count=None):
        """u_inv_t of the reference's MatGSO(b, u, u_inv_t, ...) with enable_inverse_transform (gso_interface.h:96-110,
        gso.cpp:84-158: every row operation b_i += x b_j is mirrored as u_inv_t_j -= x u_inv_t_i) for a transformation
        that started from the identity: the inverse transpose of u — an integer matrix, u being unimodular.  DERIVED ON
        THE HOST from the u the device tracks (exact integer elimination, `inverse_transpose`); the device keeps no
        u_inv_t of its own.  [count][d][d] Python-int object arrays (entries may leave int64)."""
        u = self.get_transform(first, count)
        return np.stack([inverse_transpose(m) for m in u])

    def update_gso(self):
        st = np.zeros(self.batch, dtype=np.int32)
        self._chk(self.lib.fphip_gso_update(self.h, st.ctypes.data_as(ctypes.c_void_p)), "update_gso")
        return st

    def size_reduction(self, kappa_min=0, kappa_end=-1, eta=LLL_DEF_ETA):
        st = np.zeros(self.batch, dtype=np.int32)
        self._chk(self.lib.fphip_gso_size_reduce(self.h, kappa_min, kappa_end, eta,
                                                 st.ctypes.data_as(ctypes.c_void_p)), "size_reduction")
        return st

    def lll(self, kappa_min=0, kappa_start=0, kappa_end=-1, delta=LLL_DEF_DELTA, eta=LLL_DEF_ETA, flags=0):
        """LLLReduction::lll(kappa_min, kappa_start, kappa_end) on every lattice (lll.cpp:44-164); flags are
        fplll's LLLFlags: LLL_SIEGEL (4) and LLL_EARLY_RED (2), both on the device.
        Returns (status[batch], info[batch][4] = final_kappa, n_swaps, zeros, iterations)."""
        st = np.zeros(self.batch, dtype=np.int32)
        info = np.zeros((self.batch, 4), dtype=np.int32)
        fn = self.lib.fphip_gso_lll_flags
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                       ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        rc = fn(self.h, kappa_min, kappa_start, kappa_end, delta, eta, flags,
                st.ctypes.data_as(ctypes.c_void_p), info.ctypes.data_as(ctypes.c_void_p))
        self._chk(rc, "lll")
        return st, info

    def session_lll(self, resume, kappa_min=0, kappa_start=0, kappa_end=-1, delta=LLL_DEF_DELTA, eta=LLL_DEF_ETA,
                    dirty=None, flags=0):
        """The same lll() on a RESIDENT MatGSO (fphip_gso_session_lll): resume=False starts a session from the
        basis on the device, resume=True continues it after the caller's row operations `dirty` =
        {row position: new integer row} (batch of one; with enable_transform() the row of b followed by the row
        of u, n + d integers).  Returns (status[batch], info[batch][4])."""
        st = np.zeros(self.batch, dtype=np.int32)
        info = np.zeros((self.batch, 4), dtype=np.int32)
        fn = self.lib.fphip_gso_session_lll
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                       ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.c_void_p]
        dirty = dirty or {}
        pos = np.ascontiguousarray(sorted(dirty), dtype=np.int32)
        rows = np.ascontiguousarray([dirty[p] for p in sorted(dirty)], dtype=np.int64).reshape(len(pos), -1 if len(pos) else self.n)
        assert len(pos) == 0 or rows.shape[1] in (self.n, self.n + self.d)
        self._chk(fn(self.h, 1 if resume else 0, kappa_min, kappa_start, kappa_end, delta, eta, flags, len(pos),
                     pos.ctypes.data_as(ctypes.c_void_p) if len(pos) else None,
                     rows.ctypes.data_as(ctypes.c_void_p) if len(pos) else None,
                     st.ctypes.data_as(ctypes.c_void_p), info.ctypes.data_as(ctypes.c_void_p)), "session_lll")
        return st, info

    def session_read(self, lattice=0):
        """The state the last session_lll left, in position order: (b[d][n], mu[d][d], r[d][d], valid_cols[d],
        row_expo[d]) — mu(i,j), r(i,j) are meaningful for j < valid_cols[i]."""
        b = np.zeros((self.d, self.n), dtype=np.int64)
        mu = np.zeros((self.d, self.d), dtype=np.float64)
        r = np.zeros((self.d, self.d), dtype=np.float64)
        vc = np.zeros(self.d, dtype=np.int32)
        ex = np.zeros(self.d, dtype=np.int64)
        fn = self.lib.fphip_gso_session_read
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 5
        self._chk(fn(self.h, lattice, *(a.ctypes.data_as(ctypes.c_void_p) for a in (b, mu, r, vc, ex))),
                  "session_read")
        return b, mu, r, vc, ex

    def session_read_transform(self, lattice=0):
        """u [d][d] in position order as the last session_lll left it (enable_transform() before the session)."""
        u = np.zeros((self.d, self.d), dtype=np.int64)
        fn = self.lib.fphip_gso_session_read_transform
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self._chk(fn(self.h, lattice, u.ctypes.data_as(ctypes.c_void_p)), "session_read_transform")
        return u

    def lll_ex(self, precision=106, kappa_min=0, kappa_start=0, kappa_end=-1, delta=LLL_DEF_DELTA,
               eta=LLL_DEF_ETA):
        """LLLReduction::lll in double-double (precision 106) or plain double (53) — lll_x.hip; the
        second stage of the LLL-side precision ladder.  Returns (status[batch], info[batch][4])."""
        st = np.zeros(self.batch, dtype=np.int32)
        info = np.zeros((self.batch, 4), dtype=np.int32)
        fn = self.lib.fphip_gso_lll_ex
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                       ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self._chk(fn(self.h, kappa_min, kappa_start, kappa_end, delta, eta, precision,
                     st.ctypes.data_as(ctypes.c_void_p), info.ctypes.data_as(ctypes.c_void_p)), "lll_ex")
        return st, info

    def lll_ladder(self, kappa_min=0, kappa_start=0, kappa_end=-1, delta=LLL_DEF_DELTA, eta=LLL_DEF_ETA):
        """Wrapper::lll's precision ladder on the device (wrapper.cpp:281-359): the exact double kernel,
        then double-double for the lattices that fail in double.  Returns (status, info, stage)."""
        st = np.zeros(self.batch, dtype=np.int32)
        info = np.zeros((self.batch, 4), dtype=np.int32)
        stage = np.zeros(self.batch, dtype=np.int32)
        fn = self.lib.fphip_gso_lll_ladder
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                       ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        self._chk(fn(self.h, kappa_min, kappa_start, kappa_end, delta, eta,
                     st.ctypes.data_as(ctypes.c_void_p), info.ctypes.data_as(ctypes.c_void_p),
                     stage.ctypes.data_as(ctypes.c_void_p)), "lll_ladder")
        return st, info, stage

    def _bkz_limits(self, max_time, dump_gso):
        """BKZ_MAX_TIME / BKZ_DUMP_GSO: BKZParam::max_time (seconds) and dump_gso_filename; returns the flag bits."""
        if max_time is None and dump_gso is None:
            return 0
        fn = self.lib.fphip_gso_bkz_limits
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_char_p]
        self._chk(fn(self.h, float(max_time or 0.0), None if dump_gso is None else os.fsencode(dump_gso)), "bkz_limits")
        return (0x8 if max_time is not None else 0) | (0x40 if dump_gso is not None else 0)

    def bkz(self, block_size, delta=LLL_DEF_DELTA, eta=LLL_DEF_ETA, max_loops=0, auto_abort=False, max_time=None,
            dump_gso=None):
        """BKZReduction::bkz() with empty strategies on every (LLL-reduced) lattice
        (bkz.cpp:522-668).  max_time (seconds) / dump_gso (file name): BKZ_MAX_TIME / BKZ_DUMP_GSO.
        Returns (status[batch], info[batch][4] = tours, nodes lo, nodes hi,
        enumeration calls)."""
        st = np.zeros(self.batch, dtype=np.int32)
        info = np.zeros((self.batch, 4), dtype=np.int32)
        flags = (0x4 if max_loops > 0 else 0) | (0x20 if auto_abort else 0)  # fplll's BKZFlags
        flags |= self._bkz_limits(max_time, dump_gso)
        rc = self.lib.fphip_gso_bkz(self.h, block_size, delta, eta, flags,
                                    max_loops, st.ctypes.data_as(ctypes.c_void_p),
                                    info.ctypes.data_as(ctypes.c_void_p))
        if rc == _lib.FPHIP_UNSUPPORTED:
            raise NotImplementedError("block sizes above 64 / other BKZ variants stay on the CPU")
        self._chk(rc, "bkz")
        return st, info

    def bkz_strategies(self, block_size, strategies, rnd, delta=LLL_DEF_DELTA, eta=LLL_DEF_ETA,
                       max_loops=0, gh_bnd=False, bounded_lll=False, gh_factor=1.1, auto_abort=False,
                       sd=False, handoff=False, slide=False, prune_in_loop=None, max_time=None, dump_gso=None):
        """BKZReduction::bkz() with a strategies table (preprocessing tours, pruning, GH bound,
        rerandomisation; bkz.cpp:43-124, 274-441, 522-668) on every (LLL-reduced) lattice.
        strategies: dict with the flattened arrays of include/fplll_hip.h's fphip_strategies
        (max_block_size, pre_off, pre, prune_off, prune_gh, prune_exp, coeff_off, coeff) or None.
        rnd(lattice, n) -> gmp_urandomm_ui of that lattice's generator (fplll: RandGen); a Python
        callable, the address (ctypes.c_void_p) of a C function with fphip_rand_fn's signature, or a
        tuple (address, user pointer).
        sd / slide: BKZ_SD_VARIANT / BKZ_SLD_RED (self-dual BKZ / slide reduction).
        handoff: FPHIP_BKZ_HANDOFF — large blocks are enumerated by the multi-wave enumerator (another
        visiting order than the reference's: accept the result by the reducedness predicate, not by the
        reference's basis; include/fplll_hip.h).
        prune_in_loop: FPHIP_BKZ_PRUNE_IN_LOOP — dict(preproc_cost, target, min_block, pruner_flags,
        on_device): the primal blocks of the top-level tour of at least min_block rows are pruned one by
        one by prune() on their own r-profile (on the device's volume kernel with on_device, default)
        where the reference would pick a set of the strategies (include/fplll_hip.h).
        Returns (status[batch], info[batch][4])."""
        if prune_in_loop is not None:
            il = dict(preproc_cost=1e6, target=0.5, min_block=24, pruner_flags=0x4, on_device=True)
            il.update(prune_in_loop)
            fn = self.lib.fphip_gso_bkz_inloop_pruning
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int]
            self._chk(fn(self.h, float(il["preproc_cost"]), float(il["target"]), int(il["min_block"]),
                         int(il["pruner_flags"]), 1 if il["on_device"] else 0), "bkz_inloop_pruning")

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
            st_ = Strat(int(strategies["max_block_size"]), arr("pre_off", np.int32),
                        arr("pre", np.int32), arr("prune_off", np.int32),
                        arr("prune_gh", np.float64), arr("prune_exp", np.float64),
                        arr("coeff_off", np.int32), arr("coeff", np.float64))
            keep.append(st_)
            sp = ctypes.byref(st_)
        RND = ctypes.CFUNCTYPE(ctypes.c_ulong, ctypes.c_void_p, ctypes.c_int, ctypes.c_ulong)
        rnd_user = None
        if rnd is None:
            cb = RND(0)
        elif callable(rnd):
            cb = RND(lambda _u, lattice, n: int(rnd(lattice, n)))
        elif isinstance(rnd, tuple):  # (address of a C fphip_rand_fn, its user pointer)
            cb = ctypes.cast(rnd[0], RND)
            rnd_user = rnd[1]
        else:  # the address of a C function with fphip_rand_fn's signature
            cb = ctypes.cast(rnd, RND)
        fn = self.lib.fphip_gso_bkz_strategies
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                       ctypes.c_int, ctypes.c_double, ctypes.c_void_p, RND, ctypes.c_void_p,
                       ctypes.c_void_p, ctypes.c_void_p]
        st = np.zeros(self.batch, dtype=np.int32)
        info = np.zeros((self.batch, 4), dtype=np.int32)
        flags = ((0x4 if max_loops > 0 else 0) | (0x80 if gh_bnd else 0) | (0x10 if bounded_lll else 0) |
                 (0x20 if auto_abort else 0) | (0x100 if sd else 0) | (0x1000 if handoff else 0) |
                 (0x200 if slide else 0) | (0x2000 if prune_in_loop is not None else 0))
        flags |= self._bkz_limits(max_time, dump_gso)  # BKZ_MAX_TIME / BKZ_DUMP_GSO
        rc = fn(self.h, block_size, delta, eta, flags, max_loops, gh_factor, sp, cb, rnd_user,
                st.ctypes.data_as(ctypes.c_void_p), info.ctypes.data_as(ctypes.c_void_p))
        if rc == _lib.FPHIP_UNSUPPORTED:
            raise NotImplementedError("block sizes above 64 / deeper preprocessing stay on the CPU")
        self._chk(rc, "bkz_strategies")
        return st, info

    def slide_pass(self, pass_, block_mask, block_size, strategies=None, rnd=None, delta=LLL_DEF_DELTA,
                   eta=LLL_DEF_ETA, gh_bnd=False, gh_factor=1.1):
        """fphip_gso_slide_pass: ONE pass of a slide tour (bkz.cpp:465-520) restricted to the blocks of
        block_mask — pass_ 1: primal blocks, 2: dual blocks, 3: the closing hkz of every block — with
        BKZ_BOUNDED_LLL (the blocks of a pass are independent only then).  The unit of the block-parallel
        mode (fplll_amd.distributed.slide_reduction_blocks).  Returns (status[batch], info[batch][4]);
        info[:, 0] of a primal pass = 1 when every block of the mask came out unchanged."""
        keep = []
        sp = None
        if strategies is not None:
            class Strat(ctypes.Structure):
                _fields_ = [("max_block_size", ctypes.c_int), ("pre_off", ctypes.c_void_p),
                            ("pre", ctypes.c_void_p), ("prune_off", ctypes.c_void_p),
                            ("prune_gh", ctypes.c_void_p), ("prune_exp", ctypes.c_void_p),
                            ("coeff_off", ctypes.c_void_p), ("coeff", ctypes.c_void_p)]

            def arr(key, dt):
                a = np.ascontiguousarray(strategies[key], dtype=dt)
                if a.size == 0:
                    a = np.zeros(1, dtype=dt)
                keep.append(a)
                return a.ctypes.data
            st_ = Strat(int(strategies["max_block_size"]), arr("pre_off", np.int32), arr("pre", np.int32),
                        arr("prune_off", np.int32), arr("prune_gh", np.float64), arr("prune_exp", np.float64),
                        arr("coeff_off", np.int32), arr("coeff", np.float64))
            keep.append(st_)
            sp = ctypes.byref(st_)
        RND = ctypes.CFUNCTYPE(ctypes.c_ulong, ctypes.c_void_p, ctypes.c_int, ctypes.c_ulong)
        rnd_user = None
        if rnd is None:
            cb = RND(0)
        elif callable(rnd):
            cb = RND(lambda _u, lattice, n: int(rnd(lattice, n)))
        elif isinstance(rnd, tuple):
            cb = ctypes.cast(rnd[0], RND)
            rnd_user = rnd[1]
        else:
            cb = ctypes.cast(rnd, RND)
        fn = self.lib.fphip_gso_slide_pass
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_double,
                       ctypes.c_void_p, RND, ctypes.c_void_p, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p,
                       ctypes.c_void_p]
        st = np.zeros(self.batch, dtype=np.int32)
        info = np.zeros((self.batch, 4), dtype=np.int32)
        rc = fn(self.h, block_size, delta, eta, 0x10 | (0x80 if gh_bnd else 0), gh_factor, sp, cb, rnd_user,
                int(pass_), int(block_mask), st.ctypes.data_as(ctypes.c_void_p), info.ctypes.data_as(ctypes.c_void_p))
        if rc == _lib.FPHIP_UNSUPPORTED:
            raise NotImplementedError("slide passes: blocks up to 64 rows, no last block of one row")
        self._chk(rc, "slide_pass")
        return st, info

    @staticmethod
    def slide_reduction_blocks(objects, block_size, strategies=None, rnd=None, delta=LLL_DEF_DELTA, eta=LLL_DEF_ETA,
                               max_loops=0, gh_bnd=False, gh_factor=1.1):
        """fphip_gso_slide_reduction_blocks: the block-parallel slide tour (BKZ_SLD_RED | BKZ_BOUNDED_LLL) over
        `objects` — batch-of-one MatGSOBatch objects, one per context / device, all holding the same LLL-reduced
        basis — driven by the C library's own host threads.  Returns (status, nodes, tours); every object then
        holds the result."""
        assert strategies is None and rnd is None, "strategies: use fplll_amd.distributed.slide_reduction_blocks"
        lib = objects[0].lib
        arr = (ctypes.c_void_p * len(objects))(*[o.h for o in objects])
        fn = lib.fphip_gso_slide_reduction_blocks
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                       ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_int)]
        st, nodes, tours = ctypes.c_int(0), ctypes.c_ulonglong(0), ctypes.c_int(0)
        rc = fn(arr, len(objects), block_size, delta, eta, 0x10 | (0x80 if gh_bnd else 0), max_loops, gh_factor,
                None, None, None, ctypes.byref(st), ctypes.byref(nodes), ctypes.byref(tours))
        if rc == _lib.FPHIP_UNSUPPORTED:
            raise NotImplementedError("block-parallel slide reduction needs BKZ_BOUNDED_LLL and blocks up to 64 rows")
        objects[0]._chk(rc, "slide_reduction_blocks")
        return st.value, nodes.value, tours.value

    def inloop_stats(self):
        """(prune() calls of the in-loop service, volume jobs on the device, inline on the host, launches)"""
        v = [ctypes.c_ulonglong(0) for _ in range(4)]
        fn = self.lib.fphip_gso_bkz_inloop_stats
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_ulonglong)] * 4
        fn(self.h, *[ctypes.byref(x) for x in v])
        return tuple(x.value for x in v)

    def get_mu_matrix(self, lattice=0):
        m = np.empty((self.d, self.d), dtype=np.float64)
        self._chk(self.lib.fphip_gso_get_mu(self.h, lattice, m.ctypes.data_as(ctypes.c_void_p)), "get_mu")
        return np.tril(m, -1)

    def get_r_matrix(self, lattice=0):
        m = np.empty((self.d, self.d), dtype=np.float64)
        self._chk(self.lib.fphip_gso_get_r(self.h, lattice, m.ctypes.data_as(ctypes.c_void_p)), "get_r")
        return np.tril(m)

    def row_expo(self, lattice=0):
        e = np.empty(self.d, dtype=np.int64)
        self._chk(self.lib.fphip_gso_get_row_expo(self.h, lattice, e.ctypes.data_as(ctypes.c_void_p)),
                  "get_row_expo")
        return e

    def get_mu(self, lattice, i, j):
        """mu(i,j) with the row exponents applied (gso_interface.h:694-702)."""
        e = self.row_expo(lattice)
        return float(np.ldexp(self.get_mu_matrix(lattice)[i, j], int(e[i] - e[j])))

    def get_r(self, lattice, i, j):
        e = self.row_expo(lattice)
        return float(np.ldexp(self.get_r_matrix(lattice)[i, j], int(e[i] + e[j])))

    # ---- host-side members of MatGSOInterface that BKZ callers use between reductions
    # (gso_interface.cpp:197-276; C ABI fphip_gso_util_*): evaluated on the downloaded diagonal of r
    def _diag(self, lattice):
        return np.ascontiguousarray(np.diag(self.get_r_matrix(lattice))), self.row_expo(lattice)

    def get_current_slope(self, lattice, start_row, stop_row):
        r, e = self._diag(lattice)
        return current_slope(r, e, start_row, stop_row)

    def get_log_det(self, lattice, start_row, end_row):
        r, e = self._diag(lattice)
        return log_det(r, e, start_row, end_row)

    def get_root_det(self, lattice, start_row, end_row):
        r, e = self._diag(lattice)
        return root_det(r, e, start_row, end_row)

    def get_slide_potential(self, lattice, start_row, end_row, block_size):
        r, e = self._diag(lattice)
        return slide_potential(r, e, start_row, end_row, block_size)

    def is_lll_reduced(self, lattice, delta=LLL_DEF_DELTA, eta=LLL_DEF_ETA):
        """is_lll_reduced(m, delta, eta) (lll.cpp:226-258) on the lattice's current mu / r (call update_gso
        first: the predicate reads the GSO, it does not recompute it)."""
        return is_lll_reduced(self.get_mu_matrix(lattice), self.get_r_matrix(lattice), self.row_expo(lattice),
                              delta, eta)

    @property
    def last_kernel_ms(self):
        return float(self.lib.fphip_gso_last_kernel_ms(self.h))

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "handle", None):  # (a closed context has released everything)
                self.lib.fphip_gso_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------
def _util(name, restype=ctypes.c_double):
    fn = getattr(_lib.load(), name)
    fn.restype = restype
    return fn


def _re(r_diag, row_expo):
    r = np.ascontiguousarray(r_diag, dtype=np.float64)
    e = None if row_expo is None else np.ascontiguousarray(row_expo, dtype=np.int64)
    return r, e, r.ctypes.data_as(ctypes.c_void_p), (None if e is None else e.ctypes.data_as(ctypes.c_void_p))


def current_slope(r_diag, row_expo, start_row, stop_row):
    """MatGSOInterface::get_current_slope (gso_interface.cpp:197-218) on the STORED diagonal of r and the
    row exponents (None: GSO without row exponents)."""
    r, e, rp, ep = _re(r_diag, row_expo)
    fn = _util("fphip_gso_util_current_slope")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    return fn(rp, ep, int(start_row), int(stop_row))


def log_det(r_diag, row_expo, start_row, end_row):
    """get_log_det (gso_interface.cpp:230-242)."""
    r, e, rp, ep = _re(r_diag, row_expo)
    fn = _util("fphip_gso_util_log_det")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return fn(rp, ep, r.size, int(start_row), int(end_row))


def root_det(r_diag, row_expo, start_row, end_row):
    """get_root_det (gso_interface.cpp:220-228)."""
    r, e, rp, ep = _re(r_diag, row_expo)
    fn = _util("fphip_gso_util_root_det")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return fn(rp, ep, r.size, int(start_row), int(end_row))


def slide_potential(r_diag, row_expo, start_row, end_row, block_size):
    """get_slide_potential (gso_interface.cpp:244-258)."""
    r, e, rp, ep = _re(r_diag, row_expo)
    fn = _util("fphip_gso_util_slide_potential")
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    return fn(rp, ep, r.size, int(start_row), int(end_row), int(block_size))


def adjust_radius_to_gh_bound(max_dist, max_dist_expo, block_size, root_det_, gh_factor):
    """adjust_radius_to_gh_bound (gso_interface.cpp:260-276): returns the new max_dist."""
    fn = _util("fphip_gso_util_adjust_radius_to_gh_bound")
    fn.argtypes = [ctypes.c_double, ctypes.c_long, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    return fn(float(max_dist), int(max_dist_expo), int(block_size), float(root_det_), float(gh_factor))


def is_lll_reduced(mu, r, row_expo, delta=LLL_DEF_DELTA, eta=LLL_DEF_ETA):
    """is_lll_reduced<ZT, double> (lll.cpp:226-258) on stored mu / r (d x d) and the row exponents (or None)."""
    m = np.ascontiguousarray(mu, dtype=np.float64)
    rr = np.ascontiguousarray(r, dtype=np.float64)
    e = None if row_expo is None else np.ascontiguousarray(row_expo, dtype=np.int64)
    fn = _util("fphip_gso_util_is_lll_reduced", ctypes.c_int)
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    return bool(fn(m.ctypes.data_as(ctypes.c_void_p), rr.ctypes.data_as(ctypes.c_void_p),
                   None if e is None else e.ctypes.data_as(ctypes.c_void_p), m.shape[0], float(delta), float(eta)))


def sweep_bytes_8d(d, n):
    """ALGORITHMIC bytes of one size-reduction sweep of a d x n lattice as SURVEY.md 8(d) /
    BASELINE.md 3.6 define them: per row kappa ONE babai iteration with every X_j != 0,
    B_sweep(kappa) = 8 kappa^2 [mu triangle read twice] + 8 n (2 kappa + 2) [kappa basis rows + RMW
    of row kappa, kappa float rows for the Gram row].  62.3 MB for 180 x 180.  This is the
    numerator of `roofline.achieved`."""
    return sum(8 * k * k + 8 * n * (2 * k + 2) for k in range(d))


def sweep_bytes(d, n):
    """The same plus the CONFIRMING update_gso_row the reference runs after every effective babai
    iteration (lll.cpp:172-176: the loop only ends on a pass that finds no |mu| > eta): another
    Gram row (8 n kappa) and recurrence (4 kappa^2) per row — real reference work, reported beside
    the 8(d) figure, never as `achieved`.  93.2 MB for 180 x 180."""
    return sweep_bytes_8d(d, n) + sum(4 * k * k + 8 * n * k for k in range(d))


def _unreduced_copy(b, ops_per_row=3, seed=1):
    """LLL-reduced basis → same lattice with rows that need size reduction again (every row gets a
    few multiples of earlier rows added; deterministic)."""
    rng = np.random.default_rng(seed)
    b = b.copy()
    d = b.shape[0]
    for i in range(1, d):
        for _ in range(ops_per_row):
            j = int(rng.integers(0, i))
            c = int(rng.integers(-3, 4))
            b[i] += c * b[j]
    return b


def _dense_unreduced(b, kmax=3, seed=1):
    """Size-reduced basis → the same lattice with EVERY mu(i,j), j < i, pushed out of [-1/2, 1/2]:
    row i gets c_ij * (original row j) added for every j < i, c_ij uniform in ±1..±kmax (a unit
    lower-triangular transformation; entries grow by about kmax*sqrt(d)).  One babai iteration
    with a multiplier for (nearly) every j brings it back — the all-X_j-nonzero sweep that
    SURVEY.md 8(d) prices."""
    rng = np.random.default_rng(seed)
    d = b.shape[0]
    u = rng.integers(1, kmax + 1, size=(d, d)) * rng.choice(np.array([-1, 1]), size=(d, d))
    u = np.tril(u, -1) + np.eye(d, dtype=np.int64)
    return u.astype(np.int64) @ b.astype(np.int64)


def load_basis_txt(path):
    """fplll's matrix text format ([[a b ...] [...]]), plain or gzipped; rows need not be as long as the
    matrix is high (the knapsack bases of `latticegen r` are d x (d + 1))."""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        text = f.read()
    rows = [r for r in text.replace("[[", "[").split("[") if r.strip(" ]\n\t")]
    mat = [[int(t) for t in r.replace("]", " ").split()] for r in rows]
    mat = [r for r in mat if r]
    if len({len(r) for r in mat}) != 1:
        raise ValueError("rows of different lengths in " + path)
    return np.array(mat, dtype=np.int64)


def save_basis_txt(path, b):
    """The text form Matrix<T>::print writes in compact mode (nr/matrix.cpp:136-161: `[[a b c]` newline
    `[d e f]]`) and fplll's readers (operator>>, the `fplll` command line) accept."""
    b = np.asarray(b)
    with open(path, "w") as f:
        f.write("[")
        for i, row in enumerate(b):
            f.write(("" if i == 0 else "\n") + "[" + " ".join(str(int(v)) for v in row) + "]")
        f.write("]\n")


class BKZAutoAbort:
    """fplll's BKZAutoAbort (bkz.h:33-79, bkz.cpp:800-809) over one lattice of a MatGSOBatch (or anything with
    get_current_slope(lattice, start_row, stop_row)): test_abort() is True once the slope of log r_ii has
    not improved for max_no_dec consecutive calls."""

    def __init__(self, m, num_rows, start_row=0, lattice=0):
        self.m, self.num_rows, self.start_row, self.lattice = m, int(num_rows), int(start_row), int(lattice)
        self.old_slope = float(np.finfo(np.float64).max)
        self.no_dec = -1

    def test_abort(self, scale=1.0, max_no_dec=5):
        new_slope = -self.m.get_current_slope(self.lattice, self.start_row, self.num_rows)
        if self.no_dec == -1 or new_slope < scale * self.old_slope:
            self.no_dec = 0
        else:
            self.no_dec += 1
        self.old_slope = min(self.old_slope, new_slope)
        return self.no_dec >= max_no_dec


def bench_inputs(distinct=64, kmax=3):
    """`distinct` different 180x180 inputs of the roofline measurement: the C3 basis (BKZ-20-reduced
    by the reference) under `distinct` random unit lower-triangular transformations with no zero
    below the diagonal — every row needs a multiplier for (nearly) every earlier row."""
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = load_basis_txt(os.path.join(here, "tests", "golden", "basis_q180_seed0_lll_bkz20.txt"))
    return np.stack([_dense_unreduced(base, kmax, 1000 + s) for s in range(distinct)])


def bench_roofline(ctx, batch=None, reps=3, distinct=64, traffic=None):
    """Batched size-reduction sweep on `batch` 180x180 lattices (`distinct` different ones, tiled
    over the batch on the device; each copy has its own memory), timed with HIP events on the
    launch stream; returns the `roofline` object of bench.py.  `achieved` uses the SURVEY.md 8(d)
    numerator (62.3 MB per lattice); the figure with the reference's confirming pass counted is
    reported beside it.  `traffic` (HBM bytes per launch from rocprofv3 PMC passes of this very
    workload, see bench.py) is passed in by the caller — it cannot be collected in the timed run."""
    import os
    if batch is None:
        batch = int(os.environ.get("FPHIP_GSO_BENCH_BATCH", "8192"))
    distinct = min(distinct, batch)
    bs = bench_inputs(distinct)
    d = bs.shape[1]
    g = MatGSOBatch(ctx, batch, d, d)
    try:
        times = []
        for _ in range(reps):
            g.set_basis(bs, first=0)
            g.tile_basis(distinct)
            st = g.size_reduction(0, d)
            assert int(st.min()) == 1 and int(st.max()) == 1, "size reduction failed on device"
            times.append(g.last_kernel_ms)
        # every copy of an input must have come out the same (bit for bit)
        out = g.get_basis(0, min(batch, 2 * distinct))
        for L in range(distinct, out.shape[0]):
            assert np.array_equal(out[L], out[L - distinct]), "replicas of one input differ"
        ms = float(np.mean(times))  # average launch duration
        alg = sweep_bytes_8d(d, d) * batch
        alg2 = sweep_bytes(d, d) * batch
        achieved = alg / (ms * 1e-3) / 1e9
        ver = os.environ.get("FPHIP_GSO_SWEEP", "2")
        return {
            "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
            "frac": achieved / 8000.0, "traffic": traffic,
            "kernel": ("gso_sweep2_kernel<3>" if ver != "1" else "gso_sweep_kernel<3>") +
                      " (size_reduction sweep, %d lattices of %dx%d, %d distinct inputs, a "
                      "multiplier for every j < i)" % (batch, d, d, distinct),
            "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_per_lattice": alg // batch,
            "kernel_ms": ms, "kernel_ms_each_launch": times,
            # what THIS kernel's algorithm has to move per lattice of 180 x 180 — two Gram passes over the 2-byte
            # column mirror (16.0 MB), the recurrence over the anchored mu columns (17.3), the size-reduction sweep over
            # the mu rows (8.6), the working row (0.1), at 128-byte line granularity (the table of
            # round 4; tests/perf/README.md), plus the 5.0 MB it writes: the SURVEY 8(d) numerator above prices
            # 8-byte elements and one Gram pass, so "traffic / algorithmic" flatters a kernel that moves 2-byte mirrors
            "kernel_model": ({"what": "bytes the kernel's own two-pass, 2-byte-mirror algorithm needs (line-granular)",
                              "fetch_bytes_per_lattice": 42.0e6, "write_bytes_per_lattice": 5.0e6,
                              "bytes_per_launch": 47.0e6 * batch,
                              "achieved_on_model_GBps": 47.0e6 * batch / (ms * 1e-3) / 1e9,
                              "frac_on_model": 47.0e6 * batch / (ms * 1e-3) / 8e12}
                             if (d == 180 and os.environ.get("FPHIP_GSO_NARROW", "2") == "2") else None),
            "with_confirming_pass": {
                "what": "same launch priced with the confirming update_gso_row of every row "
                        "(lll.cpp:172-176) counted as well",
                "bytes_per_lattice": alg2 // batch,
                "achieved": alg2 / (ms * 1e-3) / 1e9, "frac": alg2 / (ms * 1e-3) / 1e9 / 8000.0},
        }
    finally:
        g.close()


def smoke(ctx, C):
    """Small batched sweep vs the reference golden vector (bit-exact)."""
    import os
    f = C.load_gso_fixture(os.path.join(C.GOLDEN, "gso_q48_p3.json"))
    g = MatGSOBatch(ctx, 3, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * 3))
    st = g.size_reduction(0, f["d"])
    assert list(st) == [1, 1, 1]
    for L in range(3):
        assert np.array_equal(g.get_basis(L, 1)[0], f["b_out"])
        assert np.array_equal(g.get_mu_matrix(L), f["mu1"])
        assert np.array_equal(g.get_r_matrix(L), f["r1"])
        assert np.array_equal(g.row_expo(L), f["row_expo1"])
    print("smoke: batched size_reduction d=%d == reference (b, mu, r, row_expo bit-exact), %.3f ms"
          % (f["d"], g.last_kernel_ms))
    g.close()
