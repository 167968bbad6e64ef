"""Per-phase timers of the LLL kernel's block streams (a library built with -DFPHIP_LLL_PROF=1 must be in place:
tests/perf/build_lll_variants.sh PROF).  Usage: lll_prof.py d batch"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fplll_amd
from fplll_amd import _lib
from fplll_amd.gso import MatGSOBatch

def qary(rng, d, k, q):
    b = np.zeros((d, d), dtype=np.int64)
    b[:k, :k] = np.eye(k, dtype=np.int64)
    b[:k, k:] = rng.integers(0, q, size=(k, d - k))
    b[k:, k:] = q * np.eye(d - k, dtype=np.int64)
    return b

d = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(0)
bs = np.stack([qary(rng, d, d // 2, 1048583) for _ in range(B)])
ctx = fplll_amd.Context(0)
g = MatGSOBatch(ctx, B, d, d)
g.set_basis(bs)
lib = _lib.load()
out = (ctypes.c_uint64 * 24)()
lib.fphip_debug_lll_prof(out, 24)  # clear
st, info = g.lll()
assert np.all(st == 1)
lib.fphip_debug_lll_prof(out, 24)
v = list(out)
waves, iters, ktick = v[22], v[21], v[20]
print("shader clock (s_memtime ticks per 10 ns tick of the real-time counter): %.1f MHz" % (v[23] / ktick * 100.0))
print("d=%d B=%d: kernel %.1f ms; %d waves, %.0f iterations per wave, %.2f us per iteration"
      % (d, B, g.last_kernel_ms, waves, iters / waves, ktick * 0.01 / iters))
for k, name in enumerate(["gram", "rec", "sweep", "axpy", "single"]):
    n, rows, tf, ta = v[4 * k:4 * k + 4]
    if n:
        print("  %-6s %.2f per iteration, %.1f rows each, start-up %.2f us, total %.2f us each = %.1f %% of the kernel"
              % (name, n / iters, rows / n, tf * 0.01 / n, ta * 0.01 / n, 100.0 * ta / ktick))
g.close(); ctx.close()
