"""Host-side mirror of fplll's MatHouseholder interface (fplll/householder.h) for a BATCH of
lattices on the GPU: ``refresh_R_bf(); update_R(); get_R(expo)``.  No arithmetic here."""
import ctypes

import numpy as np

from . import _lib


def _bind(lib):
    if getattr(lib, "_hh_bound", False):
        return
    vp = ctypes.c_void_p
    lib.fphip_hh_create.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.POINTER(vp)]
    lib.fphip_hh_create.restype = ctypes.c_int
    lib.fphip_hh_destroy.argtypes = [vp]
    lib.fphip_hh_destroy.restype = None
    lib.fphip_hh_set_basis.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
    lib.fphip_hh_broadcast_basis.argtypes = [vp, ctypes.c_int]
    lib.fphip_hh_update_R.argtypes = [vp, vp]
    lib.fphip_hh_update_R_blocked.argtypes = [vp, vp]
    lib.fphip_hh_update_R_blocked.restype = ctypes.c_int
    lib.fphip_hh_hlll_ex.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                     ctypes.c_int, vp, vp]
    lib.fphip_hh_hlll_ex.restype = ctypes.c_int
    lib.fphip_hh_hlll_ladder.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                         vp, vp, vp]
    lib.fphip_hh_hlll_ladder.restype = ctypes.c_int
    lib.fphip_hh_get_R_lo.argtypes = [vp, ctypes.c_int, vp]
    lib.fphip_hh_get_R_lo.restype = ctypes.c_int
    lib.fphip_hh_get_basis.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
    lib.fphip_hh_hlll.argtypes = [vp, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                  ctypes.c_double, vp, vp]
    lib.fphip_hh_get_R.argtypes = [vp, ctypes.c_int, vp]
    lib.fphip_hh_size_reduce.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.fphip_hh_size_reduce.restype = ctypes.c_int
    lib.fphip_hh_get_row_expo.argtypes = [vp, ctypes.c_int, vp]
    lib.fphip_hh_last_kernel_ms.argtypes = [vp]
    lib.fphip_hh_last_kernel_ms.restype = ctypes.c_double
    lib._hh_bound = True


class MatHouseholderBatch:
    def __init__(self, ctx, batch, d, n, row_expo=False):
        self.ctx, self.lib = ctx, ctx.lib
        ctx.adopt(self)
        _bind(self.lib)
        self.batch, self.d, self.n = batch, d, n
        self.h = ctypes.c_void_p()
        rc = self.lib.fphip_hh_create(ctx.handle, batch, d, n, 1 if row_expo else 0,
                                      ctypes.byref(self.h))
        if rc == _lib.FPHIP_UNSUPPORTED:
            raise NotImplementedError("d, n > 256 are not handled on the device")
        if rc != _lib.FPHIP_OK:
            raise _lib.HipError("fphip_hh_create: " + ctx.last_error())

    def _chk(self, rc, what):
        if rc != _lib.FPHIP_OK:
            raise _lib.HipError("%s: %s" % (what, self.ctx.last_error()))

    def set_basis(self, b, first=0):
        b = np.ascontiguousarray(b, dtype=np.int64)
        if b.ndim == 2:
            b = b[None]
        assert b.shape[1:] == (self.d, self.n)
        self._chk(self.lib.fphip_hh_set_basis(self.h, first, b.shape[0],
                                              b.ctypes.data_as(ctypes.c_void_p)), "set_basis")

    def broadcast_basis(self, src=0):
        self._chk(self.lib.fphip_hh_broadcast_basis(self.h, src), "broadcast_basis")

    def update_R(self, blocked=False):
        """refresh_R_bf() + update_R() for every lattice.  blocked=True: the opt-in MFMA compact-WY
        mode (same R to rounding — 1e-9 on mu / r —, not bit for bit)."""
        st = np.zeros(self.batch, dtype=np.int32)
        fn = self.lib.fphip_hh_update_R_blocked if blocked else self.lib.fphip_hh_update_R
        self._chk(fn(self.h, st.ctypes.data_as(ctypes.c_void_p)), "update_R")
        return st

    def size_reduce(self, kappa, size_reduction_end=None, size_reduction_start=0):
        """MatHouseholder::size_reduce(k, size_reduction_end, size_reduction_start) (householder.cpp:402-451) on every
        lattice, on the state update_R() left.  Returns (reduced[batch], status[batch])."""
        end = kappa if size_reduction_end is None else size_reduction_end
        red = np.zeros(self.batch, dtype=np.int32)
        st = np.zeros(self.batch, dtype=np.int32)
        self._chk(self.lib.fphip_hh_size_reduce(self.h, int(kappa), int(end), int(size_reduction_start),
                                                red.ctypes.data_as(ctypes.c_void_p),
                                                st.ctypes.data_as(ctypes.c_void_p)), "size_reduce")
        return red, st

    def get_basis(self, first=0, count=1):
        b = np.empty((count, self.d, self.n), dtype=np.int64)
        self._chk(self.lib.fphip_hh_get_basis(self.h, first, count,
                                              b.ctypes.data_as(ctypes.c_void_p)), "get_basis")
        return b

    def hlll(self, delta=0.99, eta=0.51, theta=0.001, c=0.1, precision=None):
        """HLLLReduction::hlll() on every lattice (fplll/hlll.cpp:26-169).
        precision None: the exact-order double kernel (bit-identical decisions); 106: double-double
        arithmetic (FP_NR<dd_real>'s stand-in), 212: quad-double (FP_NR<qd_real>'s), 53: double — all with tree sums
        (hlll_x.hip).
        Returns (status[batch], info[batch][2] = swaps, iterations)."""
        st = np.zeros(self.batch, dtype=np.int32)
        info = np.zeros((self.batch, 2), dtype=np.int32)
        if precision is None:
            self._chk(self.lib.fphip_hh_hlll(self.h, delta, eta, theta, c,
                                             st.ctypes.data_as(ctypes.c_void_p),
                                             info.ctypes.data_as(ctypes.c_void_p)), "hlll")
        else:
            self._chk(self.lib.fphip_hh_hlll_ex(self.h, delta, eta, theta, c, int(precision),
                                                st.ctypes.data_as(ctypes.c_void_p),
                                                info.ctypes.data_as(ctypes.c_void_p)), "hlll_ex")
        return st, info

    def hlll_ladder(self, delta=0.99, eta=0.51, theta=0.001, c=0.1):
        """The wrapper's precision ladder on the device: double for the batch, double-double for the
        lattices that raise a precision alarm, quad-double for those it gives up on (wrapper.cpp:478-529, 630-710).
        Returns (status, info, stage[batch] in {53, 106, 212})."""
        st = np.zeros(self.batch, dtype=np.int32)
        info = np.zeros((self.batch, 2), dtype=np.int32)
        stage = np.zeros(self.batch, dtype=np.int32)
        self._chk(self.lib.fphip_hh_hlll_ladder(self.h, delta, eta, theta, c, st.ctypes.data_as(ctypes.c_void_p),
                                                info.ctypes.data_as(ctypes.c_void_p),
                                                stage.ctypes.data_as(ctypes.c_void_p)), "hlll_ladder")
        return st, info, stage

    def get_R_lo(self, lattice=0):
        """low plane of R after hlll(precision=106)"""
        R = np.empty((self.d, self.n))
        self._chk(self.lib.fphip_hh_get_R_lo(self.h, lattice, R.ctypes.data_as(ctypes.c_void_p)), "get_R_lo")
        return R

    def get_R(self, lattice=0):
        R = np.empty((self.d, self.n))
        self._chk(self.lib.fphip_hh_get_R(self.h, lattice, R.ctypes.data_as(ctypes.c_void_p)), "get_R")
        e = np.empty(self.d, dtype=np.int64)
        self._chk(self.lib.fphip_hh_get_row_expo(self.h, lattice, e.ctypes.data_as(ctypes.c_void_p)),
                  "get_row_expo")
        return R, e

    @property
    def last_kernel_ms(self):
        return float(self.lib.fphip_hh_last_kernel_ms(self.h))

    def close(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "handle", None):  # (a closed context has released everything)
                self.lib.fphip_hh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
