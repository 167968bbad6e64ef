#!/usr/bin/env python
"""BASELINE config 5's lattice at full size on the device, FT = double: B copies of the 256-dim
NTRU-like basis of tests/golden/c5_hlll_n256_double.json.gz, HLLLReduction::hlll on each in one
launch (fphip_hh_hlll), every result compared with the reference's reduced basis (146 491 swaps;
12 s in the reference on one core).  One wavefront runs one lattice: expect minutes per launch.
usage: hlll_c5.py [batch] [precision]"""
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
from fplll_amd.householder import MatHouseholderBatch  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
precision = int(sys.argv[2]) if len(sys.argv) > 2 else None  # 106: double-double, 53: tree-sum double
f = C.load_hlll_fixture(os.path.join(C.GOLDEN, "c5_hlll_n256_double.json.gz"))
ctx = fplll_amd.Context(0)
g = MatHouseholderBatch(ctx, batch, f["d"], f["n"], row_expo=True)
g.set_basis(np.stack([f["b_in"]] * batch))
t = time.time()
st, info = g.hlll(f["delta"], f["eta"], f["theta"], f["c"], precision=precision)
wall = time.time() - t
out = g.get_basis(0, batch)
ok = bool(np.all(st == f["status"])) and all(np.array_equal(out[L], f["b_out"]) for L in range(batch))
print(json.dumps({"config": "C5 lattice (n=256 NTRU-like), HLLL, " + ("FT=double (exact order)" if precision is None else "precision %d (hlll_x)" % precision), "batch": batch,
                  "parity_all_lattices": ok, "wall_s": wall, "kernel_s": g.last_kernel_ms / 1e3,
                  "lattices_per_s": batch / wall, "reference_s_per_lattice_1core": f["ref_seconds"],
                  "speedup_vs_reference_1core": (batch / wall) * f["ref_seconds"],
                  "swaps": int(info[0][0])}))
g.close()
ctx.close()
