"""Device self-dual BKZ (BKZ_SD_VARIANT through fphip_gso_bkz_strategies: bkzs_body<NQ, true> in
bkzs_kernel.hip) against the reference's bkzd_*sd* fixtures (oracle-pinned by
test_bkz_dual_variants_oracle_vs_ref.py): BKZ_MAX_LOOPS, the forced auto-abort (13 tours) and dual
blocks combined with strategies — all three run by default."""
import glob
import os

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu

FIXTURES = sorted(glob.glob(os.path.join(C.GOLDEN, "bkzd_*sd*.json")))


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-5])
def test_sd_bkz_matches_reference(ctx, path):
    from fplll_amd.gso import MatGSOBatch
    f = C.load_bkz_fixture(path)
    batch = 2
    g = MatGSOBatch(ctx, batch, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * batch))
    S = f.get("strategies")
    rnd, draws = C.gmp_streams_native(batch, f["rng_seed"]) if S is not None else (None, lambda: 0)
    st, info = g.bkz_strategies(f["block_size"], S, rnd, f["delta"], f["eta"],
                                max_loops=f["max_loops"], gh_bnd=bool(f["flags"] & 0x80),
                                bounded_lll=bool(f["flags"] & 0x10), gh_factor=f["gh_factor"],
                                auto_abort=bool(f["flags"] & 0x20), sd=True)
    out = g.get_basis()
    nodes = [(int(i[1]) & 0xffffffff) | ((int(i[2]) & 0xffffffff) << 32) for i in info]
    C.note(lambda: ("status", st, "expected", f["status"], "tours/calls", info[:, 0], info[:, 3], "nodes", nodes,
          "expected", f["nodes"], "kernel ms", g.last_kernel_ms, "rng draws", draws(),))
    for L in range(batch):
        bad = np.nonzero((out[L] != f["b_out"]).any(axis=1))[0]
        assert st[L] == f["status"], (L, st, info)
        assert bad.size == 0, ("first differing row", int(bad[0]), "nodes", nodes[L], f["nodes"])
        assert nodes[L] == f["nodes"]
    g.close()
