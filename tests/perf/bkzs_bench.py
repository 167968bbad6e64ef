#!/usr/bin/env python
"""Throughput of the batched device BKZ WITH strategies (fphip_gso_bkz_strategies) next to one host
core: B copies of a bkzs_* fixture lattice (BKZ-40, preprocessing tours + pruning + GH bound, two
tours) in one launch; every lattice must end on the reference's basis and node count.
usage: bkzs_bench.py [batch] [fixture-substring]"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import conftest as C  # noqa: E402
import fplll_amd  # noqa: E402
from fplll_amd.gso import MatGSOBatch  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
which = sys.argv[2] if len(sys.argv) > 2 else "pre_gh"
path = [p for p in C.bkz_strategy_fixtures() if which in p][0]
f = C.load_bkz_fixture(path)

# one host core: the C restatement of the reference (pinned bit-exact to it on this fixture)
t = time.time()
o = C.OracleGSO(f["b_in"])
o.bkz_param(f["block_size"], f["delta"], f["eta"], f["flags"], f["max_loops"], f["gh_factor"],
            f["strategies"], f["rng_seed"])
cpu_s = time.time() - t

ctx = fplll_amd.Context(0)
g = MatGSOBatch(ctx, batch, f["d"], f["n"])
g.set_basis(np.stack([f["b_in"]] * batch))
rnd, draws = C.gmp_streams_native(batch, f["rng_seed"])
t = time.time()
st, info = g.bkz_strategies(f["block_size"], f["strategies"], rnd, f["delta"], f["eta"],
                            max_loops=f["max_loops"], gh_bnd=bool(f["flags"] & 0x80),
                            bounded_lll=bool(f["flags"] & 0x10), gh_factor=f["gh_factor"])
wall = time.time() - t
out = g.get_basis()
ok = bool((st == f["status"]).all() and all(np.array_equal(out[L], f["b_out"]) for L in range(batch)))
nodes = [(int(i[1]) & 0xffffffff) | (int(i[2]) << 32) for i in info]
ok = ok and all(nd == f["nodes"] for nd in nodes)
print(json.dumps({
    "fixture": os.path.basename(path), "batch": batch, "parity_all_lattices": ok,
    "kernel_s": g.last_kernel_ms / 1e3, "wall_s": wall,
    "device_reductions_per_s": batch / wall,
    "cpu_port_s_per_reduction": cpu_s, "cpu_reductions_per_s_1core": 1.0 / cpu_s,
    "speedup_vs_1core": (batch / wall) * cpu_s,
    "enum_calls_per_lattice": int(info[0][3]), "nodes_per_lattice": nodes[0],
    "rng_draws": int(draws())}))
g.close()
ctx.close()
