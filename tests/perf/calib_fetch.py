"""FETCH_SIZE calibration: stream a known number of bytes with the sweep's load instruction."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fplll_amd
ctx = fplll_amd.Context(0)
lib = ctx.lib
lib.fphip_debug_stream.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_longlong,
                                   ctypes.POINTER(ctypes.c_double)]
for (rows, row_bytes, stride) in [(4_000_000, 2048, 2048), (4_000_000, 1440, 1440), (4_000_000, 720, 1440)]:
    ms = ctypes.c_double()
    lib.fphip_debug_stream(ctx.handle, rows, row_bytes, stride, ctypes.byref(ms))
    req = rows * ((row_bytes + 15) // 16 * 16)
    print("rows=%d row_bytes=%d stride=%d requested=%.3f GB time=%.3f ms -> %.1f GB/s" %
          (rows, row_bytes, stride, req / 1e9, ms.value, req / ms.value / 1e6), flush=True)
