"""Diagnostic: a sub-solution call on a block of 130 rows after calls on other wide blocks in the same context
(per-level counts, device vs C oracle)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import conftest as C
import fplll_amd
from fplll_amd.enumeration import FastEvaluator, enumerate_block


def run(ctx, d, seed, subs, kind):
    if kind == "cand":
        mut, rdiag, maxdist = C.wide_block_with_candidates(d, seed)
        pruning = None
    else:
        mut, rdiag, maxdist = C.wide_block(d, seed, {129: 0.15, 160: 0.2, 200: 0.3, 256: 0.3}[d])
        pruning = np.clip(np.linspace(1.0, 0.25, d)[::-1].copy(), 0.0, 1.0)[::-1].copy()
    ev, ev_o = FastEvaluator(10**9, 0), FastEvaluator(10**9, 0)
    res = enumerate_block(ctx, mut, rdiag, pruning, maxdist, ev, findsubsols=subs)
    nodes_o, _ = C.oracle_enumerate(mut, rdiag, pruning, maxdist, ev_o, findsubsols=subs)
    g = [int(v) for v in res.nodes]
    o = [int(v) for v in nodes_o]
    bad = [k for k in range(d) if g[k] != o[k]]
    print("%s d=%d subs=%d total dev %d oracle %d; levels that differ: %d %s" % (kind, d, subs, sum(g), sum(o), len(bad), bad[:6]), flush=True)


for pre in ([], [(256, 28, "plain")], [(200, 27, "plain")], [(160, 26, "plain")], [(129, 25, "plain")], [(130, 43, "cand")], [(160, 41, "cand")]):
    ctx = fplll_amd.Context(0)
    print("--- fresh context; before the sub-solution call:", pre, flush=True)
    for d, seed, kind in pre:
        run(ctx, d, seed, False, kind)
    run(ctx, 130, 43, True, "cand")
    ctx.close()
