"""Device BKZ WITH strategies (fphip_gso_bkz_strategies, bkzs_body<NQ, false> in bkzs_kernel.hip) against the reference:
tests/golden/bkzs_*.json hold BKZReduction::bkz() runs of the real reference with a strategies file
(preprocessing tours, pruning sets, GH bound, rerandomisation; see
test_bkz_strategies_oracle_vs_ref.py for the CPU-side pin of the same fixtures).  The device has to
return the reference's basis, status and enumeration node count for every lattice of the batch."""
import os

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu

# every bkzs_* fixture runs by default: the three-level nesting (bkzs_q56_b40_nested3: 24 k
# enumerations, 410 rerandomisations) and the strategies the reference's tests/test_bkz.cpp builds by
# hand (*teststrat*: preprocessing 20 -> 10 -> 5, LinearPruningParams) included.
FIXTURES = C.bkz_strategy_fixtures()
if os.environ.get("FPHIP_BKZS_ONLY"):
    FIXTURES = [p for p in FIXTURES if any(t in p for t in os.environ["FPHIP_BKZS_ONLY"].split(","))]


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-5])
def test_bkz_strategies_matches_reference(ctx, path):
    from fplll_amd.gso import MatGSOBatch
    f = C.load_bkz_fixture(path)
    batch = 3
    g = MatGSOBatch(ctx, batch, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * batch))
    # the caller's generator: natively for the long streams, through a Python callable for one
    # fixture (both forms of the mirror's `rnd` argument)
    if "r40" in path:
        py = C.GmpStreams(batch, f["rng_seed"])
        rnd, draws = py, (lambda: py.draws)
    else:
        rnd, draws = C.gmp_streams_native(batch, f["rng_seed"])
    st, info = g.bkz_strategies(f["block_size"], f["strategies"], rnd, f["delta"], f["eta"],
                                max_loops=f["max_loops"], gh_bnd=bool(f["flags"] & 0x80),
                                bounded_lll=bool(f["flags"] & 0x10), gh_factor=f["gh_factor"],
                                auto_abort=bool(f["flags"] & 0x20))
    out = g.get_basis()
    nodes = [(int(i[1]) & 0xffffffff) | (int(i[2]) << 32) for i in info]
    C.note(lambda: ("status", st, "expected", f["status"], "tours/calls", info[:, 0], info[:, 3], "nodes", nodes,
          "expected", f["nodes"], "kernel ms", g.last_kernel_ms, "rng draws", draws(),))
    for L in range(batch):
        bad = np.nonzero((out[L] != f["b_out"]).any(axis=1))[0]
        assert st[L] == f["status"], (L, st, info)
        assert bad.size == 0, ("first differing row", int(bad[0]), "of", f["d"], "nodes", nodes[L], f["nodes"])
        assert nodes[L] == f["nodes"]
    g.close()


def test_big_batch_geometry_two_waves_per_workgroup(ctx, monkeypatch):
    """What a batch above 4 lattices per CU runs on (bench leg bkz40_strategies_batch): mu rows of the block in global
    memory and, for lattices of at most 64 columns, workgroups of two waves — three of them fit a CU's LDS where one
    workgroup of four does, and bkzs_kernel<1> is built for two waves per SIMD.  Forced here on a small batch: the
    reference's basis, status and node count."""
    from fplll_amd.gso import MatGSOBatch
    path = [p for p in FIXTURES if "bkzs_q56_b36_autoabort" in p]
    if not path:
        pytest.skip("fixture filtered out")
    f = C.load_bkz_fixture(path[0])
    monkeypatch.setenv("FPHIP_GSO_WAVES_PER_BLOCK", "2")
    monkeypatch.setenv("FPHIP_BKZ_MU_LDS", "0")
    batch = 5
    g = MatGSOBatch(ctx, batch, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * batch))
    rnd, draws = C.gmp_streams_native(batch, f["rng_seed"])
    st, info = g.bkz_strategies(f["block_size"], f["strategies"], rnd, f["delta"], f["eta"],
                                max_loops=f["max_loops"], gh_bnd=bool(f["flags"] & 0x80),
                                bounded_lll=bool(f["flags"] & 0x10), gh_factor=f["gh_factor"],
                                auto_abort=bool(f["flags"] & 0x20))
    out = g.get_basis()
    for L in range(batch):
        assert st[L] == f["status"]
        assert np.array_equal(out[L], f["b_out"])
        assert ((int(info[L][1]) & 0xffffffff) | (int(info[L][2]) << 32)) == f["nodes"]
    g.close()


def _qary(rng, d, k, q):
    b = np.zeros((d, d), dtype=np.int64)
    b[:k, :k] = np.eye(k, dtype=np.int64)
    b[:k, k:] = rng.integers(0, q, size=(k, d - k))
    b[k:, k:] = q * np.eye(d - k, dtype=np.int64)
    return b


def oracle_side(bs, beta, S, flags, max_loops, seed):
    """oracle LLL then oracle BKZ-with-strategies of every lattice (its own generator each)."""
    res = []
    for b in bs:
        o = C.OracleGSO(b)
        ost, _ = o.lll()
        assert ost == 1
        o2 = C.OracleGSO(o.b)
        st, info = o2.bkz_param(beta, 0.99, 0.51, flags, max_loops, 1.1, S, seed)
        res.append((st, (int(info[1]) & 0xffffffff) | (int(info[2]) << 32), int(info[4]), o2.b.copy()))
        o.close()
        o2.close()
    return res


@pytest.mark.parametrize("d,beta,which", [(44, 34, "rerand"), (52, 36, "pre_gh")])
def test_heterogeneous_batch_vs_oracle(ctx, d, beta, which):
    """DIFFERENT lattices in one launch (every wave has its own mailbox, generator and schedule):
    device LLL + device strategy-BKZ against oracle LLL + oracle strategy-BKZ."""
    from fplll_amd.gso import MatGSOBatch
    S = C.load_bkz_fixture([p for p in C.bkz_strategy_fixtures() if which in p][0])["strategies"]
    rng = np.random.default_rng(4000 + d)
    B = 4
    bs = [_qary(rng, d, d // 2, int(rng.integers(500, 20000))) for _ in range(B)]
    flags, max_loops, seed = 0x4 | 0x80, 1, 17
    want = oracle_side(bs, beta, S, flags, max_loops, seed)
    g = MatGSOBatch(ctx, B, d, d)
    g.set_basis(np.stack(bs))
    st, _ = g.lll()
    assert np.all(st == 1)
    rnd, draws = C.gmp_streams_native(B, seed)
    st, info = g.bkz_strategies(beta, S, rnd, max_loops=max_loops, gh_bnd=True)
    out = g.get_basis(0, B)
    C.note(lambda: ("rerandomisations (oracle)", [w[2] for w in want], "rng draws", draws(), "kernel ms", g.last_kernel_ms,))
    for L in range(B):
        nodes = (int(info[L][1]) & 0xffffffff) | (int(info[L][2]) << 32)
        assert st[L] == want[L][0]
        assert nodes == want[L][1]
        assert np.array_equal(out[L], want[L][3])
    g.close()


# ---- pruning per block, in the loop (FPHIP_BKZ_PRUNE_IN_LOOP) --------------------------
INLOOP = sorted(__import__("glob").glob(os.path.join(C.GOLDEN, "bkzp_*.json")))


@pytest.mark.parametrize("on_device", [True, False], ids=["volume-kernel", "host-loop"])
@pytest.mark.parametrize("path", INLOOP, ids=lambda p: os.path.basename(p)[:-5])
def test_bkz_with_inloop_pruning_matches_reference(ctx, path, on_device):
    """tests/golden/bkzp_*.json: the REAL reference driven block by block (`ref_driver bkzfix` with
    REFDRV_INLOOP: its own svp_preprocessing / Enumeration / svp_postprocessing, and its own prune<>() on the
    block's r-profile where svp_reduction would pick a set of the strategies, bkz.cpp:325).  The device tour
    with the mailbox service pruning every such block — its searches' batches on the volume kernel, or on the
    host loop — returns the reference's basis, status and node count: every coefficient of every one of the
    90 / 42 / 459 prune() runs on the way was the reference's."""
    from fplll_amd.gso import MatGSOBatch
    f = C.load_bkz_fixture(path)
    il = f["inloop"]
    assert il["prune_failures"] == 0
    batch = 2
    g = MatGSOBatch(ctx, batch, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * batch))
    rnd, draws = C.gmp_streams_native(batch, f["rng_seed"])
    st, info = g.bkz_strategies(f["block_size"], f["strategies"], rnd, f["delta"], f["eta"],
                                max_loops=f["max_loops"], gh_bnd=bool(f["flags"] & 0x80),
                                bounded_lll=bool(f["flags"] & 0x10), gh_factor=f["gh_factor"],
                                auto_abort=bool(f["flags"] & 0x20),
                                prune_in_loop=dict(preproc_cost=il["preproc_cost"], target=il["target"],
                                                   min_block=il["min_block"], pruner_flags=il["pruner_flags"],
                                                   on_device=on_device))
    out = g.get_basis()
    nodes = [(int(i[1]) & 0xffffffff) | (int(i[2]) << 32) for i in info]
    calls, dev_jobs, host_jobs, launches = g.inloop_stats()
    C.note(lambda: ("status", st, "expected", f["status"], "nodes", nodes, "expected", f["nodes"], "kernel ms",
                    g.last_kernel_ms, "prune calls", calls, "reference", il["prune_calls"], "volume jobs dev/host",
                    dev_jobs, host_jobs, "launches", launches, "reference s", f["ref_seconds"],))
    for L in range(batch):
        bad = np.nonzero((out[L] != f["b_out"]).any(axis=1))[0]
        assert st[L] == f["status"], (L, st, info)
        assert bad.size == 0, ("first differing row", int(bad[0]), "of", f["d"], "nodes", nodes[L], f["nodes"])
        assert nodes[L] == f["nodes"]
    assert calls == batch * il["prune_calls"]
    if on_device:
        assert launches > 0 and dev_jobs > 0, "the searches' batches must run on the volume kernel"
    else:
        assert launches == 0
    g.close()


def test_handoff_service_shared_by_a_batch_of_tours(ctx, monkeypatch):
    """FPHIP_BKZ_HANDOFF on a BATCH: the blocks the schedule kernels hand to the multi-wave enumerator are served by
    a pool of workers with an enumeration context each (round 5: until then one context, one block at a time, so
    only a lone tour was fast).  A parallel enumeration with a shrinking radius is order dependent, so every lattice
    is accepted by the reference's predicate (LLL-reduced, the input's volume, a slope no worse than the sequential
    run's within 5 %, better than the input's) instead of the golden basis; the status is the golden one; with ONE worker the same holds."""
    import test_a_configs_at_size_gpu as A
    from fplll_amd.gso import MatGSOBatch
    path = [p for p in FIXTURES if "q64_b40_pre_gh" in p] or FIXTURES[:1]
    f = C.load_bkz_fixture(path[0])
    ref = A._basisstat(f["b_out"])
    inp = A._basisstat(f["b_in"])
    monkeypatch.setenv("FPHIP_BKZ_HANDOFF_NODES", "200")
    for workers in ("5", "1"):
        monkeypatch.setenv("FPHIP_BKZ_HANDOFF_WORKERS", workers)
        batch = 6
        g = MatGSOBatch(ctx, batch, f["d"], f["n"])
        g.set_basis(np.stack([f["b_in"]] * batch))
        rnd, draws = C.gmp_streams_native(batch, f["rng_seed"])
        st, info = g.bkz_strategies(f["block_size"], f["strategies"], rnd, f["delta"], f["eta"],
                                    max_loops=f["max_loops"], gh_bnd=bool(f["flags"] & 0x80),
                                    bounded_lll=bool(f["flags"] & 0x10), gh_factor=f["gh_factor"],
                                    auto_abort=bool(f["flags"] & 0x20), handoff=True)
        out = g.get_basis()
        C.note(lambda: ("hand-off batch of %d, %s worker(s): status %s, enumeration calls %s, kernel %.1f ms"
                        % (batch, workers, list(st), list(info[:, 3]), g.last_kernel_ms),))
        for L in range(batch):
            assert st[L] == f["status"]
            s = A._basisstat(out[L])
            assert s["is_lll_reduced"]
            assert abs(s["log_volume"] - inp["log_volume"]) <= 1e-9 * abs(inp["log_volume"])
            # (a rerandomising strategy: the sequential run's slope is one draw; better than the input's, and within
            #  5 % of that draw)
            assert s["slope"] > inp["slope"] and s["slope"] >= ref["slope"] - 0.05 * abs(ref["slope"]), \
                (s["slope"], ref["slope"], inp["slope"])
        g.close()
