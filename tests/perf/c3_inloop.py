"""BASELINE config 3's BKZ-60 tour on the device with hand-off (FPHIP_BKZ_HANDOFF) AND pruning per block in the
loop (FPHIP_BKZ_PRUNE_IN_LOOP: every top-level block of at least 40 rows pruned by prune() on its own r-profile,
searches' batches on the volume kernel), alone on the GPU: wall time, nodes, prune() calls, and the reference's
reducedness predicate on the output (ref_driver basisstat) next to the strategies-file tour's."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as C  # noqa: E402
import fplll_amd  # noqa: E402
import test_a_configs_at_size_gpu as A  # noqa: E402
from fplll_amd.gso import MatGSOBatch  # noqa: E402

f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "c3_bkz60_tour_strategies.json.gz"))
ctx = fplll_amd.Context(0, priority=-1)
g = MatGSOBatch(ctx, 1, f["d"], f["n"])
g.set_basis(np.stack([f["b_in"]]))
rnd, draws = C.gmp_streams_native(1, f["rng_seed"])
t = time.time()
st, info = g.bkz_strategies(f["block_size"], f["strategies"], rnd, f["delta"], f["eta"], max_loops=f["max_loops"],
                            gh_bnd=True, gh_factor=f["gh_factor"], handoff=True,
                            prune_in_loop=dict(preproc_cost=float(sys.argv[1]) if len(sys.argv) > 1 else 1e7,
                                               target=0.5, min_block=40, pruner_flags=0x4))
wall = time.time() - t
b = g.get_basis()[0]
print(json.dumps(dict(wall=wall, st=int(st[0]), nodes=A._nodes(info[0]), calls=int(info[0][3]),
                      inloop=list(g.inloop_stats()), stat=A._basisstat(b), ref_stat=A._basisstat(f["b_out"]),
                      in_stat=A._basisstat(f["b_in"]), ref_nodes=f["nodes"]), default=str))
g.close()
ctx.close()
