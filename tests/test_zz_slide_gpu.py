"""Slide reduction on the device (BKZ_SLD_RED through fphip_gso_bkz_strategies: the 0x200 frame of
bkzs_body<NQ, true> in bkzs_kernel.hip — passes of disjoint primal blocks until clean, the dual blocks
shifted by one row, the slide potential on the host between the tours, the closing hkz of every block;
fplll/bkz.cpp:465-520, 643-660) against the reference's bkzd_*slide* fixtures (oracle-pinned by
test_bkz_dual_variants_oracle_vs_ref.py): plain, with BKZ_BOUNDED_LLL, with strategies + BKZ_MAX_LOOPS, and
on an integer-relation lattice with a ragged last block — basis, status and node count identical."""
import glob
import os

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu

FIXTURES = sorted(glob.glob(os.path.join(C.GOLDEN, "bkzd_*slide*.json")))


@pytest.mark.parametrize("path", FIXTURES, ids=lambda p: os.path.basename(p)[:-5])
def test_slide_reduction_matches_reference(ctx, path):
    from fplll_amd.gso import MatGSOBatch
    f = C.load_bkz_fixture(path)
    assert f["flags"] & 0x200
    batch = 2
    g = MatGSOBatch(ctx, batch, f["d"], f["n"])
    g.set_basis(np.stack([f["b_in"]] * batch))
    S = f.get("strategies")
    rnd, draws = C.gmp_streams_native(batch, f["rng_seed"]) if S is not None else (None, lambda: 0)
    st, info = g.bkz_strategies(f["block_size"], S, rnd, f["delta"], f["eta"],
                                max_loops=f["max_loops"], gh_bnd=bool(f["flags"] & 0x80),
                                bounded_lll=bool(f["flags"] & 0x10), gh_factor=f["gh_factor"],
                                auto_abort=bool(f["flags"] & 0x20), slide=True)
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


# ---- block-parallel mode: the blocks of every pass over several contexts / ranks -------
def _slide_blocks_run(world, f, results, gather, rank):
    import fplll_amd
    from fplll_amd.distributed import slide_reduction_blocks
    from fplll_amd.gso import MatGSOBatch
    try:
        ctx = fplll_amd.Context(0)
        g = MatGSOBatch(ctx, 1, f["d"], f["n"])
        g.set_basis(f["b_in"][None])
        results[rank] = slide_reduction_blocks(g, rank, world, gather, f["block_size"], max_loops=f["max_loops"],
                                               delta=f["delta"], eta=f["eta"])
        g.close()
        ctx.close()
    except BaseException as e:  # noqa: reported by the caller
        results[rank] = e
        try:
            gather.bar.abort()
        except Exception:
            pass


def _potential_and_predicate(ctx, b, bs):
    """slide potential of a basis (device GSO + the reference's formula) and the reference's is_lll_reduced."""
    from fplll_amd.gso import MatGSOBatch
    g = MatGSOBatch(ctx, 1, b.shape[0], b.shape[1])
    g.set_basis(b[None])
    assert int(g.update_gso()[0]) == 1
    pot = g.get_slide_potential(0, 0, b.shape[0], bs)
    red = g.is_lll_reduced(0)
    ld = g.get_log_det(0, 0, b.shape[0])
    g.close()
    return pot, red, ld


def test_block_parallel_slide_reduction_over_contexts(ctx):
    """fplll_amd.distributed.slide_reduction_blocks with 1, 2 and 3 contexts of this GPU (one host thread
    each; every block of a pass reduced from the pass-start basis by the context it is dealt to, rows
    gathered, merged basis to everybody): the SAME basis, node count and number of tours whatever the number
    of contexts — and an output the reference's predicates accept: LLL-reduced (lll.cpp:226-258), the
    input's lattice volume, a slide potential (gso_interface.cpp:244-258) below the input's and within 0.1 %
    of the sequential reference run's (bkzd_q64_b16_slide_bounded_lll.json)."""
    import threading
    from fplll_amd.distributed import LocalGather
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "bkzd_q64_b16_slide_bounded_lll.json"))
    assert f["flags"] & 0x210 == 0x210
    runs = {}
    for world in (1, 2, 3):
        gather = LocalGather(world)
        res = [None] * world
        ts = [threading.Thread(target=_slide_blocks_run, args=(world, f, res, gather, r)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(600)
        for r in res:
            assert not isinstance(r, BaseException), repr(r)
        for r in range(1, world):
            assert res[r][0] == res[0][0] and res[r][2] == res[0][2] and np.array_equal(res[r][1], res[0][1])
        runs[world] = res[0]
    st, out, nodes, tours = runs[1]
    C.note(lambda: ("block-parallel slide: status", st, "nodes", nodes, "tours", tours, "(sequential reference:",
                    f["nodes"], "nodes)",))
    for world in (2, 3):
        assert runs[world][0] == st and runs[world][2] == nodes and runs[world][3] == tours
        assert np.array_equal(runs[world][1], out), "the result depends on the number of contexts"
    assert st == 1 and tours >= 1 and nodes > 0
    p_in, _, ld_in = _potential_and_predicate(ctx, f["b_in"], f["block_size"])
    p_out, red_out, ld_out = _potential_and_predicate(ctx, out, f["block_size"])
    p_ref, red_ref, ld_ref = _potential_and_predicate(ctx, f["b_out"], f["block_size"])
    C.note(lambda: ("slide potential: input %.6f, block-parallel %.6f, sequential reference %.6f" % (p_in, p_out, p_ref),))
    assert red_out and red_ref
    assert abs(ld_out - ld_in) < 1e-9 * abs(ld_in) and abs(ld_ref - ld_in) < 1e-9 * abs(ld_in)
    assert p_out < p_in and p_out <= p_ref + 1e-3 * abs(p_ref)


def _slide_rank(rank, world, port, path, q):
    import torch.distributed as dist
    import fplll_amd
    from fplll_amd.distributed import DistGather, slide_reduction_blocks
    from fplll_amd.gso import MatGSOBatch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = C.load_bkz_fixture(path)
    ctx = fplll_amd.Context(0)
    g = MatGSOBatch(ctx, 1, f["d"], f["n"])
    g.set_basis(f["b_in"][None])
    st, out, nodes, tours = slide_reduction_blocks(g, rank, world, DistGather(dist), f["block_size"],
                                                   max_loops=f["max_loops"], delta=f["delta"], eta=f["eta"])
    q.put((rank, st, out.tolist(), nodes, tours))
    dist.barrier()
    g.close()
    ctx.close()
    dist.destroy_process_group()


def test_block_parallel_slide_reduction_over_ranks(ctx):
    """The same over two RANKS (one process each, gloo, both on this GPU): one all_gather of the blocks'
    rows per pass; both ranks end on the basis the single-participant run ends on."""
    import socket
    import torch.multiprocessing as mp
    from fplll_amd.distributed import LocalGather
    path = os.path.join(C.GOLDEN, "bkzd_q64_b16_slide_bounded_lll.json")
    f = C.load_bkz_fixture(path)
    res = [None]
    _slide_blocks_run(1, f, res, LocalGather(1), 0)
    assert not isinstance(res[0], BaseException), repr(res[0])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    ps = [mpctx.Process(target=_slide_rank, args=(r, 2, port, path, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=600) for _ in range(2))
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    for r in range(2):
        assert out[r][1] == res[0][0] and out[r][3] == res[0][2] and out[r][4] == res[0][3]
        assert np.array_equal(np.array(out[r][2]), res[0][1])


def test_block_parallel_slide_tour_behind_the_c_abi():
    """fphip_gso_slide_reduction_blocks — the same tour driven by the C library's own host threads over 1 and 3
    contexts (what a C++ caller without Python uses): the basis, node count and number of tours of the Python
    orchestration (`slide_reduction_blocks` with LocalGather)."""
    import fplll_amd
    from fplll_amd.distributed import LocalGather
    from fplll_amd.gso import MatGSOBatch
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "bkzd_q64_b16_slide_bounded_lll.json"))
    res = [None]
    _slide_blocks_run(1, f, res, LocalGather(1), 0)
    assert not isinstance(res[0], BaseException), repr(res[0])
    st0, out0, nodes0, tours0 = res[0]
    for count in (1, 3):
        ctxs = [fplll_amd.Context(0) for _ in range(count)]
        gs = [MatGSOBatch(c, 1, f["d"], f["n"]) for c in ctxs]
        for g in gs:
            g.set_basis(f["b_in"][None])
        st, nodes, tours = MatGSOBatch.slide_reduction_blocks(gs, f["block_size"], delta=f["delta"], eta=f["eta"],
                                                              max_loops=f["max_loops"])
        outs = [g.get_basis(0, 1)[0] for g in gs]
        for g in gs:
            g.close()
        for c in ctxs:
            c.close()
        assert (st, nodes, tours) == (st0, nodes0, tours0), (count, st, nodes, tours, st0, nodes0, tours0)
        for o in outs:
            assert np.array_equal(o, out0), "the C ABI tour differs from the Python orchestration (count %d)" % count
