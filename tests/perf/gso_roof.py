import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fplll_amd
from fplll_amd import gso as G
ctx = fplll_amd.Context(0)
for B in [int(a) for a in sys.argv[1:]] or [4096]:
    r = G.bench_roofline(ctx, batch=B, reps=2)
    print("B=%d kernel_ms=%.2f achieved=%.1f GB/s frac=%.3f" % (B, r["kernel_ms"], r["achieved"], r["frac"]), flush=True)
