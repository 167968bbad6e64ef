"""First-contact GPU probe: runs a few fixtures through the C ABI with verbose diagnostics."""
import os, sys, time, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import conftest as C
import fplll_amd
from fplll_amd.enumeration import FastEvaluator, enumerate_block

ctx = fplll_amd.Context(0)
for path in C.enum_fixtures():
    f = C.load_fixture(path)
    ev = FastEvaluator(f["max_sols"], f["strategy"])
    t = time.time()
    res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev)
    dt = time.time() - t
    ok_nodes = [int(v) for v in res.nodes] == f["nodes"]
    st = res.stats
    print("%-28s nodes %10d ref %10d eq=%s final=%s ref=%s | wall %.2f ms kern %.3f ms (final %.3f) phases %d tasks %d L %d ovf %d sols %d"
          % (f["name"], res.total_nodes, f["total_nodes"], ok_nodes, res.final_maxdist == f["final_maxdist"],
             f["final_maxdist"], dt * 1e3, st.kernel_ms, st.final_kernel_ms, st.phases, st.final_tasks,
             st.final_root_level, st.overflowed, st.solutions), flush=True)

# larger synthetic trees (fixed bound → exact counts known from the oracle sizing runs)
def lin(d, c):
    return np.maximum(0.05, 1.0 - c * np.arange(d) / d)
for (d, seed, slope, rf, c, expect) in [(64, 7, 0.055, 1.02, 1.25, 11261041), (64, 7, 0.055, 1.02, 1.15, 1049078970)]:
    if len(sys.argv) > 1 and sys.argv[1] == "small" and expect > 10**8:
        continue
    mut, rdiag, maxdist = C.synthetic_block(d, seed, slope, rf)
    for rep in range(2):
        ev = FastEvaluator(10**9, 0)
        t = time.time()
        res = enumerate_block(ctx, mut, rdiag, lin(d, c), maxdist, ev)
        dt = time.time() - t
        st = res.stats
        print("synthetic d=%d c=%.2f nodes %d expect %d eq=%s wall %.2f ms kern %.2f ms launches %d tasks %d L %d ovf %d -> %.3e nodes/s"
              % (d, c, res.total_nodes, expect, res.total_nodes == expect, dt * 1e3, st.kernel_ms, st.phases,
                 st.final_tasks, st.final_root_level, st.overflowed, res.total_nodes / dt), flush=True)
