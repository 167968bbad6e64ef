#!/usr/bin/env python
"""BASELINE config 3 on the device, batched: B copies of the 180-dim q-ary lattice, ONE BKZ-60 tour
with the pruner strategies (BKZ_MAX_LOOPS 1, BKZ_GH_BND) through fphip_gso_bkz_strategies, every
result compared with the reference's golden (tests/golden/c3_bkz60_tour_strategies.json.gz: 15 160
enumerations, 1.22e9 nodes, 11 rerandomisations per lattice; 49 s in the reference on one core).
One wavefront runs one lattice (expect minutes per launch): the figure of merit is lattices/s of
the batch.   usage: bkzs_c3.py [batch]"""
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

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "c3_bkz60_tour_strategies.json.gz"))
ctx = fplll_amd.Context(0)
g = MatGSOBatch(ctx, batch, f["d"], f["n"])
g.set_basis(np.stack([f["b_in"]] * batch))
rnd, draws = C.gmp_streams_native(batch, f["rng_seed"])
t = time.time()
st, info = g.bkz_strategies(f["block_size"], f["strategies"], rnd, f["delta"], f["eta"],
                            max_loops=f["max_loops"], gh_bnd=True, gh_factor=f["gh_factor"])
wall = time.time() - t
out = g.get_basis()
nodes = [(int(i[1]) & 0xffffffff) | ((int(i[2]) & 0xffffffff) << 32) for i in info]
ok = bool((st == f["status"]).all()) and all(nd == f["nodes"] for nd in nodes) and \
    all(np.array_equal(out[L], f["b_out"]) for L in range(batch))
print(json.dumps({"config": "C3: n=180 q-ary, BKZ-60 one tour, pruner strategies, GH bound 1.1",
                  "batch": batch, "parity_all_lattices": ok, "status": [int(x) for x in st[:4]],
                  "wall_s": wall, "kernel_s": g.last_kernel_ms / 1e3,
                  "tours_per_s": batch / wall, "reference_s_per_tour_1core": f["ref_seconds"],
                  "speedup_vs_reference_1core": (batch / wall) * f["ref_seconds"],
                  "nodes_per_lattice": nodes[0], "expected_nodes": f["nodes"],
                  "rng_draws": int(draws())}))
g.close()
ctx.close()
