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
                            exchange_chunks=4, **kw):
    """One SVP enumeration over all ranks of `dist` (one process per GPU): this rank walks its share of
    the subtree tasks (fplll_amd.enumeration.enumerate_block with the 16-byte bound exchange), then
    the reductions above.  Returns (dist, coords, nodes, local_result): the first three identical on
    every rank.  `evaluator` is this rank's (fplll's evaluator is per process); with BEST-1 semantics
    its shortest solution is this rank's candidate."""
    from .enumeration import enumerate_block
    rank, world = dist.get_rank(), dist.get_world_size()
    res = enumerate_block(ctx, mut, rdiag, pruning, maxdist, evaluator, shard_index=rank, shard_count=world,
                          exchange=make_exchange(dist, device), exchange_chunks=exchange_chunks, **kw)
    sols = sorted(evaluator.solutions, key=lambda s: s[0]) if evaluator.solutions else []
    bd = sols[0][0] if sols else float("inf")
    bc = sols[0][1] if sols else None
    gd, gc, gn = reduce_enumeration(dist, bd, bc, res.nodes, len(rdiag), device)
    return gd, gc, gn, res
