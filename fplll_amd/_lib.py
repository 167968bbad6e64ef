"""ctypes binding of the C ABI declared in include/fplll_hip.h (libfplll_hip.so).

The product path has no CPU fallback: if the HIP library is missing, or no GPU is visible when a
context is created, this module raises.  (Loading the library and listing its symbols works
without a GPU; that is what the CPU-side ABI test checks.)
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfplll_hip.so")

FPHIP_OK = 0
FPHIP_UNSUPPORTED = 1
FPHIP_ERROR = -1
ENUM_MAX_DIM = 256

SOL_CB = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p, ctypes.c_double,
                          ctypes.POINTER(ctypes.c_double))
SUBSOL_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_double,
                             ctypes.POINTER(ctypes.c_double), ctypes.c_int)
EXCHANGE_CB = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p, ctypes.c_double, ctypes.c_int,
                               ctypes.POINTER(ctypes.c_int))


GATHER_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                             ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t))


class EnumOpts(ctypes.Structure):
    _fields_ = [
        ("dual", ctypes.c_int),
        ("findsubsols", ctypes.c_int),
        ("shard_index", ctypes.c_int),
        ("shard_count", ctypes.c_int),
        ("exchange", EXCHANGE_CB),
        ("exchange_user", ctypes.c_void_p),
        ("exchange_chunks", ctypes.c_int),
        ("target_tasks", ctypes.c_int),
        ("phase_growth", ctypes.c_int),
        ("waves_per_block", ctypes.c_int),
        ("min_nodes_decline", ctypes.c_int),
        ("gather", GATHER_CB),
        ("gather_user", ctypes.c_void_p),
    ]


class EnumStats(ctypes.Structure):
    _fields_ = [
        ("total_nodes", ctypes.c_uint64),
        ("solutions", ctypes.c_uint64),
        ("wall_ms", ctypes.c_double),
        ("kernel_ms", ctypes.c_double),
        ("final_kernel_ms", ctypes.c_double),
        ("phases", ctypes.c_int),
        ("final_tasks", ctypes.c_int),
        ("final_root_level", ctypes.c_int),
        ("overflowed", ctypes.c_int),
        ("bfs_restarts", ctypes.c_int),
        ("moved_tasks", ctypes.c_uint64),
    ]


_lib = None


def load():
    """Load libfplll_hip.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "fplll_amd: %s is missing — run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp = ctypes.c_void_p
    lib.fphip_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    lib.fphip_create.restype = ctypes.c_int
    lib.fphip_destroy.argtypes = [vp]
    lib.fphip_destroy.restype = None
    lib.fphip_last_error.argtypes = [vp]
    lib.fphip_last_error.restype = ctypes.c_char_p
    lib.fphip_device_count.restype = ctypes.c_int
    lib.fphip_abi_version.restype = ctypes.c_int
    lib.fphip_enum_run.argtypes = [
        vp, ctypes.c_int, ctypes.c_double, vp, vp, vp, ctypes.POINTER(EnumOpts), SOL_CB, SUBSOL_CB,
        vp, vp, ctypes.POINTER(EnumStats)
    ]
    lib.fphip_enum_run.restype = ctypes.c_int
    _lib = lib
    return lib


class HipError(RuntimeError):
    pass


class Context:
    """One context per GPU (one process per GPU; device = LOCAL_RANK)."""

    def __init__(self, device=0, priority=0):
        """priority: stream priority class of the context, -1 low / 0 normal / +1 high (fphip_create_ex): a
        context that runs minutes-long single launches beside others should be low."""
        import weakref
        self.lib = load()
        self._children = weakref.WeakSet()  # batch objects living on this context
        self.handle = ctypes.c_void_p()
        self.lib.fphip_create_ex.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        self.lib.fphip_create_ex.restype = ctypes.c_int
        rc = self.lib.fphip_create_ex(int(device), int(priority), ctypes.byref(self.handle))
        if rc != FPHIP_OK:
            msg = self.lib.fphip_last_error(self.handle).decode() if self.handle else "?"
            if self.handle:
                self.lib.fphip_destroy(self.handle)
                self.handle = None
            raise HipError("fphip_create(device=%d) failed: %s" % (device, msg))

    def last_error(self):
        return self.lib.fphip_last_error(self.handle).decode()

    def adopt(self, child):
        """Batch objects register here: they hold device memory of this context and must be
        released before it (a test that fails half-way leaves them to the garbage collector)."""
        self._children.add(child)

    def close(self):
        if getattr(self, "handle", None):
            for c in list(getattr(self, "_children", ())):
                try:
                    c.close()
                except Exception:
                    pass
            self.lib.fphip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
