import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import conftest as C
import fplll_amd
from fplll_amd.enumeration import FastEvaluator, enumerate_block
ctx = fplll_amd.Context(0)
def lin(d, c):
    return np.maximum(0.05, 1.0 - c * np.arange(d) / d)
which = sys.argv[1] if len(sys.argv) > 1 else "mid"
cases = {"mid": (64, 7, 0.055, 1.02, 1.25, 11261041), "big": (64, 7, 0.055, 1.02, 1.15, 1049078970)}
d, seed, slope, rf, c, expect = cases[which]
mut, rdiag, maxdist = C.synthetic_block(d, seed, slope, rf)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    ev = FastEvaluator(10**9, 0)
    t = time.time()
    res = enumerate_block(ctx, mut, rdiag, lin(d, c), maxdist, ev)
    dt = time.time() - t
    st = res.stats
    print("d=%d c=%.2f nodes %d eq=%s wall %.2f ms kern %.2f ms launches %d tasks %d L %d ovf %d -> %.3e nodes/s"
          % (d, c, res.total_nodes, res.total_nodes == expect, dt * 1e3, st.kernel_ms, st.phases, st.final_tasks,
             st.final_root_level, st.overflowed, res.total_nodes / dt), flush=True)
