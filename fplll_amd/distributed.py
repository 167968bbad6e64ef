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


def task_shard(prefix, root_level, d, shard_count):
    """The content hash enum_kernel.hip uses to assign a subtree task to a rank: computed from the
    coefficient prefix x[root_level..d) only, so every rank derives the same owner whatever the
    order of the task in its buffer.  (Python restatement for the CPU tests.)"""
    h = 0
    for lane in range(root_level, d):
        x = int(prefix[lane]) & 0xFFFFFFFF
        h = (h + x * ((2654435761 * (lane + 1)) & 0xFFFFFFFF)) & 0xFFFFFFFF
    h ^= h >> 15
    return h % shard_count


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
