"""Per-shard work of one C3 enumeration (fixed bound so that the tree is the same for every shard):
proxy for multi-GPU strong-scaling efficiency = mean/max of per-shard kernel time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import conftest as C
import fplll_amd
from fplll_amd.enumeration import FastEvaluator, enumerate_block
ctx = fplll_amd.Context(0)
name = sys.argv[1] if len(sys.argv) > 1 else "c3_b60_k2_linear30_input"
f = C.load_fixture(os.path.join(C.GOLDEN, name + ".json"))
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.9
for world in [int(a) for a in sys.argv[3:]] or [1, 2, 8]:
    ms, nodes = [], []
    for s in range(world):
        ev = FastEvaluator(10**9, 0)   # fixed bound: every shard sees the same tree
        res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"] * scale, ev,
                              shard_index=s, shard_count=world, exchange_chunks=1)
        ms.append(res.stats.kernel_ms); nodes.append(res.total_nodes)
    print("world=%d total nodes %d  kernel ms per shard: max %.1f mean %.1f -> balance %.2f ; nodes max/mean %.2f"
          % (world, sum(nodes), max(ms), np.mean(ms), np.mean(ms) / max(ms), max(nodes) / np.mean(nodes)), flush=True)
