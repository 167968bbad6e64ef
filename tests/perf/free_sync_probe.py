"""Does hipFree / the library's create-destroy cycle wait for kernels of OTHER streams?  (It decides
whether the long device runs of the GPU suite can overlap with the rest of it in one process.)"""
import os, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
import numpy as np
import conftest as C
import fplll_amd
from fplll_amd.gso import MatGSOBatch
from fplll_amd.householder import MatHouseholderBatch

f = C.load_hlll_fixture(os.path.join(C.GOLDEN, "c5_hlll_n256_double.json.gz"))
done = {}
def long_run():
    c = fplll_amd.Context(0)
    h = MatHouseholderBatch(c, 1, 256, 256, row_expo=True)
    h.set_basis(f["b_in"][None])
    t = time.time(); h.hlll(precision=53); done["t"] = time.time() - t
    h.close(); c.close()
th = threading.Thread(target=long_run); th.start()
time.sleep(2.0)
ctx = fplll_amd.Context(0)
g0 = C.load_gso_fixture(os.path.join(C.GOLDEN, "gso_q48_p3.json"))
for i in range(5):
    t = time.time()
    g = MatGSOBatch(ctx, 4, g0["d"], g0["n"]); g.set_basis(np.stack([g0["b_in"]] * 4))
    t1 = time.time(); g.size_reduction(); t2 = time.time(); g.close(); t3 = time.time()
    print("cycle %d: create+upload %.3f s, sweep %.3f s, destroy %.3f s (long run alive: %s)"
          % (i, t1 - t, t2 - t1, t3 - t2, th.is_alive()), flush=True)
# a strategy-BKZ call (pinned mailbox: hipHostMalloc / hipHostFree) and an enumeration beside the long run
fb = C.load_bkz_fixture([p for p in C.bkz_strategy_fixtures() if "pre_gh" in p][0])
for i in range(2):
    t = time.time()
    g = MatGSOBatch(ctx, 2, fb["d"], fb["n"]); g.set_basis(np.stack([fb["b_in"]] * 2))
    rnd = C.GmpStreams(2, fb["rng_seed"])
    st, info = g.bkz_strategies(fb["block_size"], fb["strategies"], rnd, fb["delta"], fb["eta"], max_loops=fb["max_loops"],
                                gh_bnd=bool(fb["flags"] & 0x80), gh_factor=fb["gh_factor"])
    g.close()
    print("bkz_strategies cycle %d: %.3f s, kernel %.3f s (long run alive: %s)" % (i, time.time() - t, g.last_kernel_ms / 1e3 if False else 0.0, th.is_alive()), flush=True)
from fplll_amd.enumeration import FastEvaluator, enumerate_block
fe = C.load_fixture(os.path.join(C.GOLDEN, "enum_d40_lin20_fixed.json"))
t = time.time(); enumerate_block(ctx, fe["mut"], fe["rdiag"], fe["pruning"], fe["maxdist"], FastEvaluator(fe["max_sols"], fe["strategy"]))
print("enumeration: %.3f s (long run alive: %s)" % (time.time() - t, th.is_alive()), flush=True)
th.join(); print("long run took %.1f s" % done["t"])
