"""Multi-GPU glue: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in
the CPU tests).  The enumeration path shards by subtree; its only exchange is the best bound
(and whether any rank still has subtree tasks), a 16-byte all-reduce at chunk / round boundaries
(SURVEY.md §8(e)).  This module holds that collective and the deterministic partition rule so
that both can be tested without a GPU.
"""
import numpy as np


def make_exchange(dist, device="cpu"):
    """Return ``exchange(local_bound, local_active) -> (global_bound, any_active)``.

    One all_reduce(MIN) over the pair (bound, -active): MIN of the bounds, and -1 if any rank is
    still active.  Every rank must call it the same number of times (fphip_exchange_cb contract).
    """
    import torch
    buf = torch.zeros(2, dtype=torch.float64, device=device)

    def exchange(local_bound, local_active):
        buf[0] = float(local_bound)
        buf[1] = -1.0 if local_active else 0.0
        dist.all_reduce(buf, op=dist.ReduceOp.MIN)
        v = buf.tolist()
        return v[0], v[1] < 0.0

    return exchange


def make_gather(dist, device="cpu"):
    """Return ``gather(block: bytes) -> list of every rank's block`` (rank order): what fphip_gather_cb needs — the
    all-gather the work movement between ranks rides on (counts of donated tasks, then the surplus tasks, at the
    round boundaries of the walk).  Two collectives: the sizes, then the blocks padded to the largest.  Every
    rank must call it the same number of times."""
    import torch
    world = dist.get_world_size()

    def gather(block):
        n = torch.tensor([len(block)], dtype=torch.int64, device=device)
        ns = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(ns, n)
        sizes = [int(t.item()) for t in ns]
        m = max(sizes)
        if m == 0:
            return [b""] * world
        buf = torch.zeros(m, dtype=torch.uint8, device=device)
        if len(block):
            buf[: len(block)] = torch.frombuffer(bytearray(block), dtype=torch.uint8).to(device)
        out = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(out, buf)
        return [bytes(out[r][: sizes[r]].cpu().numpy().tobytes()) for r in range(world)]

    return gather


def balance_plan(counts, room=None, fraction=16):
    """The plan rebalance_tasks (enum_host.hip) derives from the ranks' numbers of donated tasks — restated for
    the CPU tests: targets differ by at most one, ranks above theirs give the surplus (the tail of their list),
    ranks below take consecutive slices of the pooled surpluses in rank order.  Returns (moved, surplus[],
    deficit[], offset[]) with offset[r] = where rank r's slice of the pool starts; moved = 0 when the lists are
    level enough (less than a sixteenth of the tasks would move; `fraction` = FPHIP_MOVE_FRACTION) or when a
    receiver has no room for what it would get (`room[r]`: free rows of rank r's table of level-64 ancestors —
    blocks above 64 rows, whose tasks travel with their ancestor's row; the ranks gather it with the counts)."""
    W = len(counts)
    total = sum(counts)
    target = [total // W + (1 if r < total % W else 0) for r in range(W)]
    surplus = [max(0, c - t) for c, t in zip(counts, target)]
    deficit = [max(0, t - c) for c, t in zip(counts, target)]
    moved = sum(surplus)
    if moved == 0 or moved * fraction < total:
        return 0, [0] * W, [0] * W, [0] * W
    if room is not None and any(dd > rr for dd, rr in zip(deficit, room)):
        return 0, [0] * W, [0] * W, [0] * W
    offset = [sum(deficit[:r]) for r in range(W)]
    return moved, surplus, deficit, offset


def task_key(prefix, root_level, d):
    """64-bit content key of a subtree task (task_key_kernel in enum_kernel.hip): computed from the
    coefficient prefix x[root_level..d) only, so every rank derives the same key whatever the
    position of the task in its buffer.  (Python restatement for the CPU tests.)"""
    M = 0xFFFFFFFF
    h1 = h2 = 0
    for lane in range(root_level, d):
        x = int(prefix[lane]) & M
        h1 = (h1 + x * ((2654435761 * (lane + 1)) & M)) & M
        h2 = (h2 + (x ^ 0x9E3779B9) * ((40503 * (2 * lane + 3) + 2246822519) & M)) & M
    return (h1 << 32) | h2


def partition_tasks(partdists, keys, shard_count):
    """The rule fphip_enum_run uses to deal subtree tasks to ranks (enum_host.hip, first walk
    round): sort by (partial distance of the root ascending = heaviest subtree first, content key),
    then deal the sorted list in snake order 0..W-1,W-1..0.  Returns, per rank, the list of task
    indices in walking order.  Depends on task CONTENT only."""
    order = sorted(range(len(keys)), key=lambda i: (partdists[i], keys[i]))
    W = shard_count
    shares = [[] for _ in range(W)]
    for p, i in enumerate(order):
        r = p % (2 * W)
        shares[r if r < W else 2 * W - 1 - r].append(i)
    return shares


def shard_batch(batch, rank, world):
    """Replicas only: GSO / LLL / HLLL / BKZ of ONE lattice do not shard, a batch
    of independent lattices does, with no data-path collective — rank r takes the contiguous slice
    [lo, hi) of the batch (sizes differ by at most one).  Returns (lo, hi)."""
    base, rem = divmod(int(batch), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_status(dist, local_status, batch, rank, world, device="cpu"):
    """Collect the per-lattice status words of every rank's slice on all ranks (one all_gather of
    `batch` int32 at the end of a batched reduction — the only communication of that path)."""
    import torch
    lo, hi = shard_batch(batch, rank, world)
    width = -(-batch // world)
    buf = torch.full((width,), -99, dtype=torch.int32, device=device)
    buf[: hi - lo] = torch.as_tensor(local_status, dtype=torch.int32).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = []
    for r in range(world):
        a, b = shard_batch(batch, r, world)
        res.extend(out[r][: b - a].tolist())
    return res


def run_rounds(exchange, rounds_local, bound0):
    """Host-side round protocol of fphip_enum_run's walk phase, for the CPU tests: a rank keeps
    calling ``exchange`` once per round until NO rank has tasks, adopting the smallest bound.
    ``rounds_local`` = list of (tasks_left_after_round, bound_found_in_round or None)."""
    bound = bound0
    i = 0
    calls = 0
    active = True
    others = True
    while active or others:
        if i < len(rounds_local):
            left, found = rounds_local[i]
            if found is not None and found < bound:
                bound = found
            active = left > 0
        else:
            active = False
        i += 1
        bound, any_active = exchange(bound, active)
        calls += 1
        others = any_active
    return bound, calls


def reduce_enumeration(dist, best_dist, best_coords, nodes, dim, device="cpu"):
    """The three reductions that turn the ranks' shares of ONE sharded enumeration into the result of
    the call: norm MIN -> the winner's coefficient vector broadcast -> per-level node
    counts SUM.  All messages are below 4 KB (latency-bound: xGMI bandwidth plays no role).

      best_dist   this rank's shortest squared norm (float; inf if it holds no solution)
      best_coords its coefficient vector (sequence of dim numbers, ignored when best_dist is inf)
      nodes       its per-level node counts (dim + 1 integers)

    Returns (dist, coords, nodes) — identical on every rank: the globally shortest norm, the vector of
    the LOWEST rank that holds it (a deterministic tie rule), and the summed counts.  One all_gather of
    (norm, vector) = 8 (dim + 1) bytes per rank and one all_reduce of dim + 1 int64."""
    import math
    import torch
    world = dist.get_world_size()
    mine = torch.zeros(dim + 1, dtype=torch.float64, device=device)
    mine[0] = float(best_dist)
    if math.isfinite(float(best_dist)):
        mine[1:] = torch.as_tensor([float(v) for v in best_coords][:dim], dtype=torch.float64)
    everyone = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    norms = [float(t[0]) for t in everyone]
    win = min(range(world), key=lambda r: (norms[r], r))
    cnt = torch.as_tensor([int(v) for v in nodes][:dim + 1], dtype=torch.int64).to(device)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    coords = [float(v) for v in everyone[win][1:]] if math.isfinite(norms[win]) else None
    return norms[win], coords, [int(v) for v in cnt.tolist()]


def enumerate_block_sharded(ctx, dist, mut, rdiag, pruning, maxdist, evaluator, device="cpu",
                            exchange_chunks=4, move_work=True, **kw):
    """One SVP enumeration over all ranks of `dist` (one process per GPU): this rank walks its share of
    the subtree tasks (fplll_amd.enumeration.enumerate_block with the 16-byte bound exchange and, at the round boundaries, the
    levelling of the donated subtrees over the ranks), then
    the reductions above.  Returns (dist, coords, nodes, local_result): the first three identical on
    every rank.  `evaluator` is this rank's (fplll's evaluator is per process); with BEST-1 semantics
    its shortest solution is this rank's candidate."""
    from .enumeration import enumerate_block
    rank, world = dist.get_rank(), dist.get_world_size()
    # (move_work: donated subtrees are levelled over the ranks at every round boundary — fphip_gather_cb)
    res = enumerate_block(ctx, mut, rdiag, pruning, maxdist, evaluator, shard_index=rank, shard_count=world,
                          exchange=make_exchange(dist, device), exchange_chunks=exchange_chunks,
                          gather=make_gather(dist, device) if (move_work and world > 1) else None, **kw)
    sols = sorted(evaluator.solutions, key=lambda s: s[0]) if evaluator.solutions else []
    bd = sols[0][0] if sols else float("inf")
    bc = sols[0][1] if sols else None
    gd, gc, gn = reduce_enumeration(dist, bd, bc, res.nodes, len(rdiag), device)
    return gd, gc, gn, res


# ---------------------------------------------------------------------------------------------------------
# Slide reduction, block-parallel (include/fplll_hip.h: fphip_gso_slide_pass)
# ---------------------------------------------------------------------------------------------------------
def slide_blocks(d, block_size):
    """(primal, dual) block lists of one slide tour over rows [0, d) (bkz.cpp:468-499): primal block i =
    rows [i bs, min(d, (i+1) bs)), dual block i = rows [i bs + 1, (i+1) bs + 1) for i < p - 1."""
    p = (d + block_size - 1) // block_size
    primal = [(i * block_size, min(d, (i + 1) * block_size)) for i in range(p)]
    dual = [(i * block_size + 1, (i + 1) * block_size + 1) for i in range(p - 1)]
    return primal, dual


def deal_blocks(nblocks, rank, world):
    """Block i of a pass belongs to participant i % world."""
    return [i for i in range(nblocks) if i % world == rank]


class LocalGather:
    """gather for participants that live in ONE process (one context per device, or several contexts on
    one device): a plain rendezvous of host arrays.  Every participant calls gather() once per pass with
    {block index: (rows, clean, nodes)}; all get the union."""

    def __init__(self, world):
        import threading
        self.world = world
        self.lock = threading.Lock()
        self.bar = threading.Barrier(world)
        self.box = {}

    def gather(self, rank, mine):
        with self.lock:
            self.box.update(mine)
        self.bar.wait()
        out = dict(self.box)
        self.bar.wait()
        if rank == 0:
            self.box.clear()
        self.bar.wait()
        return out


class DistGather:
    """gather over torch.distributed ranks (one process per GPU): one all_gather per pass of the blocks'
    rows — beta x n int64 per block, 86 KB at BASELINE config 3 — plus a (clean, nodes) pair per block."""

    def __init__(self, dist, device="cpu"):
        self.dist, self.device = dist, device

    def gather(self, rank, mine, layout):
        import torch
        world = self.dist.get_world_size()
        # fixed-size slots: every rank sends the same shape (blocks it does not own stay zero)
        width = max(hi - lo for lo, hi in layout)
        n = None
        for v in mine.values():
            n = v[0].shape[1]
        n_t = torch.tensor([n or 0], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(n_t, op=self.dist.ReduceOp.MAX)
        n = int(n_t.item())
        buf = torch.zeros((len(layout), width * n + 2), dtype=torch.int64, device=self.device)
        for i, (rows, clean, nodes) in mine.items():
            flat = torch.as_tensor(np.ascontiguousarray(rows, dtype=np.int64).reshape(-1))
            buf[i, :flat.numel()] = flat.to(self.device)
            buf[i, -2] = int(bool(clean))
            buf[i, -1] = int(nodes)
        every = [torch.empty_like(buf) for _ in range(world)]
        self.dist.all_gather(every, buf)
        out = {}
        for i, (lo, hi) in enumerate(layout):
            t = every[i % world][i].cpu().numpy()
            out[i] = (t[:(hi - lo) * n].reshape(hi - lo, n).copy(), bool(t[-2]), int(t[-1]))
        return out


def slide_reduction_blocks(g, rank, world, gather, block_size, max_loops=0, strategies=None, rnd=None,
                           delta=0.99, eta=0.51, gh_bnd=False, gh_factor=1.1):
    """BKZ_SLD_RED | BKZ_BOUNDED_LLL with the blocks of every pass dealt over `world` participants
    (devices of one process, or ranks): slide_tour (bkz.cpp:465-520) and the loop of bkz() around it
    (:567-571, 583-617, 643-660), host side.  `g` is this participant's MatGSOBatch (batch 1) holding the
    LLL-reduced input — the same on every participant.

    Per pass every participant reduces ITS blocks (block i -> participant i % world), EACH FROM THE
    PASS-START BASIS (one fphip_gso_slide_pass launch per block on a fresh copy of that basis), then the
    blocks' rows are gathered and the merged basis — block i's rows from the participant that reduced it,
    a unimodular block-triangular transformation of the pass-start basis — replaces every participant's
    copy.  The bounded LLL after a primal pass, the potential test and the closing hkz run on the merged
    basis on every participant alike (deterministic: no communication).  Because no block ever sees
    another block's new rows inside a pass, the result is independent of `world`.

    Returns (status, basis[d][n], total enumeration nodes, tours)."""
    from .gso import slide_potential
    assert g.batch == 1
    d = g.d
    # the C entry points' rules (fphip_gso_slide_reduction_blocks): a caller generator would be consumed
    # differently by every participant — the rerandomisations of the closing hkz, and with them the result,
    # would depend on `world` — and a pass has at most 64 blocks (64-bit block masks)
    if rnd is not None:
        raise ValueError("slide_reduction_blocks: the block-parallel tour is defined without rerandomisation (rnd=None)")
    if (d + block_size - 1) // block_size > 64:
        raise ValueError("slide_reduction_blocks: more than 64 blocks per pass (d=%d, block_size=%d)" % (d, block_size))
    primal, dual = slide_blocks(d, block_size)
    is_dist = isinstance(gather, DistGather)

    def run_pass(pass_, layout):
        start = g.get_basis(0, 1)[0]
        mine = {}
        for i in deal_blocks(len(layout), rank, world):
            g.set_basis(start[None])
            st, info = g.slide_pass(pass_, 1 << i, block_size, strategies, rnd, delta, eta, gh_bnd, gh_factor)
            if int(st[0]) <= 0:
                raise RuntimeError("slide pass %d block %d failed with status %d" % (pass_, i, int(st[0])))
            lo, hi = layout[i]
            nodes = (int(info[0][1]) & 0xffffffff) | ((int(info[0][2]) & 0xffffffff) << 32)
            mine[i] = (g.get_basis(0, 1)[0][lo:hi].copy(), bool(info[0][0]) if pass_ == 1 else True, nodes)
        every = gather.gather(rank, mine, layout) if is_dist else gather.gather(rank, mine)
        merged = start.copy()
        clean, nodes = True, 0
        for i, (lo, hi) in enumerate(layout):
            rows, c, nd = every[i]
            merged[lo:hi] = rows
            clean &= c
            nodes += nd
        g.set_basis(merged[None])
        return clean, nodes

    def potential():
        assert int(g.update_gso()[0]) == 1
        return g.get_slide_potential(0, 0, d, block_size)

    total_nodes, tours, status = 0, 0, 1
    old = potential()
    while True:
        if max_loops > 0 and tours >= max_loops:
            status = 8  # RED_BKZ_LOOPS_LIMIT
            break
        while True:  # primal passes until one leaves every block (and the bounded LLL) unchanged
            clean, nd = run_pass(1, primal)
            total_nodes += nd
            st, info = g.lll(0, 0, d, delta, eta)
            if int(st[0]) != 1:
                return int(st[0]), g.get_basis(0, 1)[0], total_nodes, tours
            if int(info[0][1]) > 0:
                clean = False
            if clean:
                break
        if dual:
            _, nd = run_pass(2, dual)
            total_nodes += nd
        tours += 1
        new = potential()
        if new >= old or block_size >= d:
            break
        old = new
    # closing hkz of every block (bkz.cpp:643-660) on every participant alike; counted once
    st, info = g.slide_pass(3, 0, block_size, strategies, rnd, delta, eta, gh_bnd, gh_factor)
    total_nodes += (int(info[0][1]) & 0xffffffff) | ((int(info[0][2]) & 0xffffffff) << 32)
    if int(st[0]) <= 0:
        status = int(st[0])
    return status, g.get_basis(0, 1)[0], total_nodes, tours
