"""Host protocol of block-parallel slide reduction (fplll_amd.distributed.slide_reduction_blocks; SURVEY
8(e) row 2) without a GPU: the block layout of a slide tour (fplll/bkz.cpp:468-499), the dealing of blocks
to participants, the merge of the gathered rows, the repeat-until-clean of the primal passes, the potential
test — with a toy integer "reduction" in place of the device (a MatGSOBatch look-alike whose slide_pass
sorts the rows of a block by length).  The result must not depend on the number of participants, in one
process (threads + LocalGather) and over gloo ranks (DistGather, world size 2)."""
import os
import socket
import threading

import numpy as np
import pytest

import conftest as C  # noqa: F401


class ToyBatch:
    """What slide_reduction_blocks uses of MatGSOBatch.  'Reducing' a block = sorting its rows by squared
    length (a unimodular transformation that only touches the block's rows), clean when already sorted."""
    batch = 1

    def __init__(self, b):
        self.b = np.array(b, dtype=np.int64)
        self.d, self.n = self.b.shape
        self.launches = 0

    def get_basis(self, first=0, count=None):
        return self.b[None].copy()

    def set_basis(self, b):
        self.b = np.array(b[0], dtype=np.int64)

    def slide_pass(self, pass_, mask, block_size, *a, **k):
        from fplll_amd.distributed import slide_blocks
        self.launches += 1
        primal, dual = slide_blocks(self.d, block_size)
        clean = True
        if pass_ in (1, 2):
            for i, (lo, hi) in enumerate(primal if pass_ == 1 else dual):
                if (mask >> i) & 1:
                    rows = self.b[lo:hi]
                    order = np.argsort((rows.astype(object) ** 2).sum(axis=1).astype(np.float64), kind="stable")
                    if pass_ == 2:
                        order = order[::-1]
                    clean &= bool(np.all(order == np.arange(hi - lo)))
                    self.b[lo:hi] = rows[order]
        info = np.zeros((1, 4), dtype=np.int32)
        info[0][0] = 1 if clean else 0
        info[0][1] = 7  # "nodes" of this launch
        return np.array([8 if pass_ != 3 else 1], dtype=np.int32), info

    def lll(self, *a, **k):
        return np.array([1], dtype=np.int32), np.zeros((1, 4), dtype=np.int32)

    def update_gso(self):
        return np.array([1], dtype=np.int32)

    def get_slide_potential(self, lattice, lo, hi, bs):
        # strictly decreasing over the first tours, then flat: three tours in all
        self.calls = getattr(self, "calls", 0) + 1
        return float(max(0, 3 - self.calls))


def _basis(seed, d=22, n=9):
    return np.random.default_rng(seed).integers(-50, 50, size=(d, n))


def _run_local(world, b, bs):
    from fplll_amd.distributed import LocalGather, slide_reduction_blocks
    gather = LocalGather(world)
    res = [None] * world

    def work(r):
        res[r] = slide_reduction_blocks(ToyBatch(b), r, world, gather, bs)
    ts = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    return res


def test_block_layout_of_a_slide_tour():
    from fplll_amd.distributed import deal_blocks, slide_blocks
    primal, dual = slide_blocks(64, 16)
    assert primal == [(0, 16), (16, 32), (32, 48), (48, 64)] and dual == [(1, 17), (17, 33), (33, 49)]
    primal, dual = slide_blocks(30, 8)  # ragged last block (bkz.cpp:478: min(max_row - kappa, block_size))
    assert primal[-1] == (24, 30) and len(primal) == 4 and dual == [(1, 9), (9, 17), (17, 25)]
    assert deal_blocks(5, 1, 2) == [1, 3] and sorted(deal_blocks(5, 0, 3) + deal_blocks(5, 1, 3) + deal_blocks(5, 2, 3)) == list(range(5))


@pytest.mark.parametrize("bs", [5, 8])
def test_result_does_not_depend_on_the_number_of_participants(bs):
    b = _basis(3)
    one = _run_local(1, b, bs)[0]
    assert one[0] == 1 and one[3] == 3  # three tours, then the potential stops falling
    for world in (2, 3):
        res = _run_local(world, b, bs)
        for r in range(world):
            assert res[r][0] == one[0] and res[r][2] == one[2] and res[r][3] == one[3]
            assert np.array_equal(res[r][1], one[1])
    # the rows are a permutation of the input's inside the lattice: same multiset of rows
    assert sorted(map(tuple, one[1])) == sorted(map(tuple, b))


def _rank(rank, world, port, b, bs, q):
    import torch.distributed as dist
    from fplll_amd.distributed import DistGather, slide_reduction_blocks
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    st, out, nodes, tours = slide_reduction_blocks(ToyBatch(b), rank, world, DistGather(dist), bs)
    q.put((rank, st, out.tolist(), nodes, tours))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_over_gloo_agree_with_the_single_participant():
    import torch.multiprocessing as mp
    b, bs = _basis(5), 6
    one = _run_local(1, b, bs)[0]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank, args=(r, 2, port, b, bs, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for r in range(2):
        assert out[r][1] == one[0] and out[r][3] == one[2] and out[r][4] == one[3]
        assert np.array_equal(np.array(out[r][2]), one[1])


def test_block_parallel_tour_rejects_what_would_depend_on_the_participants():
    """A caller generator would be drawn from differently by every participant (the closing hkz's
    rerandomisations would depend on `world`), and a pass with more than 64 blocks does not fit the 64-bit
    block mask — `1 << i` would be truncated to 0 and the pass would reduce nothing while reporting the block
    clean.  Both are refused up front, here and in the C entry points (fphip_gso_slide_pass,
    fphip_gso_slide_reduction_blocks)."""
    from fplll_amd.distributed import LocalGather, slide_reduction_blocks
    with pytest.raises(ValueError, match="rnd"):
        slide_reduction_blocks(ToyBatch(_basis(1)), 0, 1, LocalGather(1), 5, rnd=lambda *a: 0)
    with pytest.raises(ValueError, match="64 blocks"):
        slide_reduction_blocks(ToyBatch(_basis(2, d=180, n=4)), 0, 1, LocalGather(1), 2)
    import ctypes
    from fplll_amd import _lib
    src = open(os.path.join(os.path.dirname(_lib.LIB_PATH), "..", "csrc", "gso_host.hip")).read()
    body = src[src.index('extern "C" int fphip_gso_slide_reduction_blocks'):]
    body = body[:body.index("\n}\n")]
    assert "if (rnd)\n    return FPHIP_UNSUPPORTED;" in body and "if (p > 64)\n    return FPHIP_UNSUPPORTED;" in body
    assert ctypes  # (the library itself is exercised on the GPU: tests/test_zz_slide_gpu.py)
