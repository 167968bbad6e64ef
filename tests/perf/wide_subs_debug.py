"""Diagnostic: repeated sub-solution calls on the 130-row block with candidates; per-level count differences."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest as C
import fplll_amd
from fplll_amd.enumeration import FastEvaluator, enumerate_block

ctx = fplll_amd.Context(0)
mut, rdiag, maxdist = C.wide_block_with_candidates(130, 43)
ev_o = FastEvaluator(10**9, 0)
nodes_o, _ = C.oracle_enumerate(mut, rdiag, None, maxdist, ev_o, findsubsols=True)
o = [int(v) for v in nodes_o]
import time
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    for subs in (False, True):
        ev = FastEvaluator(10**9, 0)
        slow = {"n": 0}
        if subs and rep % 2 == 1:
            orig = ev.eval_sub_sol
            def slow_sub(offset, coord, dist, orig=orig):
                time.sleep(0.002)  # a slow consumer of the sub-solution ring
                return orig(offset, coord, dist)
            ev.eval_sub_sol = slow_sub
        res = enumerate_block(ctx, mut, rdiag, None, maxdist, ev, findsubsols=subs)
        g = [int(v) for v in res.nodes]
        diff = [(k, g[k] - o[k]) for k in range(130) if g[k] != o[k]]
        print("rep %d subs=%d slow=%d: %d levels differ %s ... %s; kernel %.1f ms phases %d" % (rep, subs, int(subs and rep % 2 == 1), len(diff), diff[:6], diff[-4:], res.stats.kernel_ms, res.stats.phases), flush=True)
ctx.close()
