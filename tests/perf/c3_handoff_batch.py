"""BASELINE config 3's BKZ-60 tour on the device in hand-off mode for a BATCH of tours (the hand-off service shared by
the batch: one worker thread and one enumeration context per worker, FPHIP_BKZ_HANDOFF_WORKERS): wall time, tours/s,
and the reference's reducedness predicate on every output (ref_driver basisstat).
usage: c3_handoff_batch.py [batch] [lattices to check with the predicate]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import conftest as C  # noqa: E402
import fplll_amd  # noqa: E402
import test_a_configs_at_size_gpu as A  # noqa: E402
from fplll_amd.gso import MatGSOBatch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ncheck = int(sys.argv[2]) if len(sys.argv) > 2 else 4
f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "c3_bkz60_tour_strategies.json.gz"))
ctx = fplll_amd.Context(0, priority=-1)
g = MatGSOBatch(ctx, B, f["d"], f["n"])
g.set_basis(np.stack([f["b_in"]] * B))
rnd, draws = C.gmp_streams_native(B, f["rng_seed"])
t = time.time()
st, info = g.bkz_strategies(f["block_size"], f["strategies"], rnd, f["delta"], f["eta"],
                            max_loops=f["max_loops"], gh_bnd=True, gh_factor=f["gh_factor"], handoff=True)
wall = time.time() - t
bs = g.get_basis()
ref = A._basisstat(f["b_out"])
inp = A._basisstat(f["b_in"])
ok = []
for L in list(range(min(ncheck, B))):
    s = A._basisstat(bs[L])
    ok.append(bool(s["is_lll_reduced"] and abs(s["log_volume"] - inp["log_volume"]) < 1e-6 * abs(inp["log_volume"])
                   and s["slope"] > inp["slope"] and abs(s["slope"] - ref["slope"]) < 0.002))
print(json.dumps({"batch": B, "wall_s": wall, "tours_per_s": B / wall, "status": sorted(set(int(x) for x in st)),
                  "expect_status": f["status"], "nodes_mean": float(np.mean([A._nodes(i) for i in info])),
                  "enum_calls_mean": float(np.mean([int(i[3]) for i in info])), "predicate_ok": ok,
                  "reference_tour_s_1core": f["ref_seconds"], "speedup_vs_1core": B / wall * f["ref_seconds"],
                  "workers": os.environ.get("FPHIP_BKZ_HANDOFF_WORKERS", "8 (default, capped by the batch and 16)")}))
g.close()
ctx.close()
