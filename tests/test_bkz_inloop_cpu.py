"""In-loop pruning (the product's FPHIP_BKZ_PRUNE_IN_LOOP) pinned WITHOUT a GPU: the C
restatement of BKZReduction::bkz with strategies (oracle/gso_oracle.c: oracle_gso_bkz_param, itself pinned to
the reference by tests/test_bkz_strategies_oracle_vs_ref.py) with its in-loop hook handing every top-level
block to the PRODUCT's pruner (fplll_amd/csrc/pruner_search.hip, host volume engine) must end on the basis,
status and node count of tests/golden/bkzp_*.json — runs of the REAL reference driven block by block with its
own prune<>() at bkz.cpp:325 (oracle/ref_driver.cpp: InloopBKZ).  Every coefficient of every prune() on the
way has to be the reference's for that; the device run of the same fixtures is tests/test_bkzs_gpu.py."""
import glob
import os

import numpy as np
import pytest

import conftest as C

FIXTURES = sorted(glob.glob(os.path.join(C.GOLDEN, "bkzp_*.json")))


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-5])
def test_oracle_schedule_with_product_pruner_matches_driven_reference(path):
    f = C.load_bkz_fixture(path)
    il = f["inloop"]
    o = C.OracleGSO(f["b_in"])
    st, info, calls = o.bkz_param_inloop(f["block_size"], f["delta"], f["eta"], f["flags"], f["max_loops"],
                                         f["gh_factor"], f["strategies"], f["rng_seed"], il)
    nodes = (int(info[1]) & 0xffffffff) | ((int(info[2]) & 0xffffffff) << 32)
    out = o.b.copy()
    o.close()
    assert calls == il["prune_calls"], "the hook must fire exactly where the driven reference pruned"
    assert st == f["status"]
    bad = np.nonzero((out != f["b_out"]).any(axis=1))[0]
    assert bad.size == 0, ("first differing row", int(bad[0]), "nodes", nodes, f["nodes"])
    assert nodes == f["nodes"]
