"""Diagnostic: per-level counts of sub-solution calls on blocks around 128 rows, device vs C oracle."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest as C
import fplll_amd
from fplll_amd.enumeration import FastEvaluator, enumerate_block

ctx = fplll_amd.Context(0)
for d, seed in ((100, 43), (128, 43), (130, 43)):
    mut, rdiag, maxdist = C.wide_block_with_candidates(d, seed) if d > 128 else C.synthetic_block(d, 23, 0.03, 0.22 if d == 100 else 0.08)
    pruning = None if d > 128 else np.linspace(1.0, 0.25, d)
    for subs in (False, True):
        ev, ev_o = FastEvaluator(10**9, 0), FastEvaluator(10**9, 0)
        res = enumerate_block(ctx, mut, rdiag, pruning, maxdist, ev, findsubsols=subs)
        nodes_o, _ = C.oracle_enumerate(mut, rdiag, pruning, maxdist, ev_o, findsubsols=subs)
        g = [int(v) for v in res.nodes]
        o = [int(v) for v in nodes_o]
        bad = [k for k in range(d) if g[k] != o[k]]
        print("d=%d subs=%d total dev %d oracle %d; levels that differ: %s" % (d, subs, sum(g), sum(o), bad[:8] + (["..."] if len(bad) > 8 else [])), flush=True)
        if bad:
            print("   dev   ", g[:8], g[60:68], g[126:132])
            print("   oracle", o[:8], o[60:68], o[126:132])
ctx.close()
