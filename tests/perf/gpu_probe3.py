import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import conftest as C
import fplll_amd
from fplll_amd.enumeration import FastEvaluator, enumerate_block
ctx = fplll_amd.Context(0)
names = sys.argv[1:] or ["c3_b60_k0_pruner", "c3_b60_k1_pruner", "c3_b60_k2_pruner"]
for name in names:
    f = C.load_fixture(os.path.join(C.GOLDEN, name + ".json"))
    for rep in range(3):
        ev = FastEvaluator(f["max_sols"], f["strategy"])
        t = time.time()
        res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev)
        dt = time.time() - t
        st = res.stats
        print("%s rep%d nodes %d (ref %d) final %.6f (ref %.6f) wall %.3f ms kern %.3f ms launches %d tasks %d L %d sols %d -> %.3e nodes/s"
              % (name, rep, res.total_nodes, f["total_nodes"], res.final_maxdist, f["final_maxdist"], dt * 1e3, st.kernel_ms,
                 st.phases, st.final_tasks, st.final_root_level, st.solutions, res.total_nodes / dt), flush=True)
