"""Device BKZ WITH strategies (fphip_gso_bkz_strategies, bkzs_kernel.hip) against the reference:
tests/golden/bkzs_*.json hold BKZReduction::bkz() runs of the real reference with a strategies file
(preprocessing tours, pruning sets, GH bound, rerandomisation; see
test_bkz_strategies_oracle_vs_ref.py for the CPU-side pin of the same fixtures).  The device has to
return the reference's basis, status and enumeration node count for every lattice of the batch."""
import os

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu

# BKZ_AUTO_ABORT is not offered by the strategies entry point yet
FIXTURES = [p for p in C.bkz_strategy_fixtures() if "autoabort" not in p]
if os.environ.get("FPHIP_BKZS_ONLY"):
    FIXTURES = [p for p in FIXTURES if any(t in p for t in os.environ["FPHIP_BKZS_ONLY"].split(","))]


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-5])
def test_bkz_strategies_matches_reference(ctx, path):
    from fplll_amd.gso import MatGSOBatch
    f = C.load_bkz_fixture(path)
    batch = 3
    g = MatGSOBatch(ctx, batch, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * batch))
    rnd = C.GmpStreams(batch, f["rng_seed"])
    st, info = g.bkz_strategies(f["block_size"], f["strategies"], rnd, f["delta"], f["eta"],
                                max_loops=f["max_loops"], gh_bnd=bool(f["flags"] & 0x80),
                                bounded_lll=bool(f["flags"] & 0x10), gh_factor=f["gh_factor"])
    out = g.get_basis()
    nodes = [(int(i[1]) & 0xffffffff) | (int(i[2]) << 32) for i in info]
    print("status", st, "expected", f["status"], "tours/calls", info[:, 0], info[:, 3], "nodes", nodes,
          "expected", f["nodes"], "kernel ms", g.last_kernel_ms, "rng draws", rnd.draws)
    for L in range(batch):
        bad = np.nonzero((out[L] != f["b_out"]).any(axis=1))[0]
        assert st[L] == f["status"], (L, st, info)
        assert bad.size == 0, ("first differing row", int(bad[0]), "of", f["d"], "nodes", nodes[L], f["nodes"])
        assert nodes[L] == f["nodes"]
    g.close()
