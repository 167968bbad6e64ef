"""World-size-2 tests of the multi-GPU glue on CPU (gloo): the bound/active all-reduce, the
termination consensus of the round protocol (every rank makes the same number of collective
calls even when their round counts differ) and the content partition (disjoint + complete,
independent of task order)."""
import os
import socket

import numpy as np
import pytest

import conftest as C


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from fplll_amd.distributed import make_exchange, run_rounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ex = make_exchange(dist, "cpu")
    # 1. plain collective
    b, a = ex(1.0 + rank, rank == 1)
    assert b == 1.0 and a is True
    b, a = ex(5.0 - rank, False)
    assert b == 4.0 and a is False
    # 2. round protocol: rank 0 has 5 rounds of work, rank 1 only 2; rank 1 finds the best bound
    rounds = {0: [(10, None), (8, 0.9), (5, None), (1, None), (0, None)],
              1: [(3, 0.7), (0, None)]}[rank]
    bound, calls = run_rounds(ex, rounds, 1.0)
    q.put((rank, bound, calls))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_and_round_consensus_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(2))
    assert res[0][1] == res[1][1] == 0.7  # both ranks end with the global best bound
    assert res[0][2] == res[1][2] == 5    # same number of collective calls on every rank


def test_content_partition_is_order_independent():
    from fplll_amd.distributed import task_key, partition_tasks
    rng = np.random.default_rng(0)
    d, L = 60, 35
    tasks = [rng.integers(-3, 4, size=d) for _ in range(5000)]
    pds = rng.random(5000).round(2)   # many exact ties: the key must break them
    keys = [task_key(t, L, d) for t in tasks]
    assert len(set(keys)) == len(keys)
    # coefficients below the root level do not matter
    t2 = tasks[0].copy()
    t2[:L] = 99
    assert task_key(t2, L, d) == keys[0]
    for world in (2, 4, 8):
        shares = partition_tasks(pds, keys, world)
        flat = sorted(i for s in shares for i in s)
        assert flat == list(range(len(tasks)))                   # complete and disjoint
        assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1
        # weight proxy (remaining radius) is balanced to well under 1%
        w = [sum(1.0 - pds[i] for i in s) for s in shares]
        assert max(w) / (sum(w) / world) < 1.01
        # every share is walked heaviest first
        for s in shares:
            assert all(pds[s[i]] <= pds[s[i + 1]] for i in range(len(s) - 1))
        # a permuted task buffer gives every task the same owner
        perm = rng.permutation(len(tasks))
        shares_p = partition_tasks([pds[i] for i in perm], [keys[i] for i in perm], world)
        for a, b in zip(shares, shares_p):
            assert a == [int(perm[j]) for j in b]


def _shard_worker(rank, world, port, q):
    import torch.distributed as dist
    from fplll_amd.distributed import gather_status, shard_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = 11
    lo, hi = shard_batch(batch, rank, world)
    local = [1000 * rank + i for i in range(lo, hi)]  # stands for the status words of the slice
    q.put((rank, lo, hi, gather_status(dist, local, batch, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_batch_sharding_replicas_world2():
    """GSO / LLL / HLLL / BKZ of a batch: contiguous slices per rank, one all_gather of statuses."""
    import torch.multiprocessing as mp
    from fplll_amd.distributed import shard_batch
    for batch in (1, 7, 8, 4096):
        for world in (1, 2, 3, 8):
            cuts = [shard_batch(batch, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == batch
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert (res[0][1], res[0][2], res[1][1], res[1][2]) == (0, 6, 6, 11)
    expect = list(range(0, 6)) + [1000 + i for i in range(6, 11)]
    assert res[0][3] == expect and res[1][3] == expect


def _reduce_worker(rank, world, port, q):
    import torch.distributed as dist
    from fplll_amd.distributed import reduce_enumeration
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dim = 6
    # case 1: rank 1 holds the shortest vector; rank 2 (if any) holds nothing
    cand = {0: (3.5, [1, 0, -2, 0, 0, 1]), 1: (2.25, [0, 1, 1, -1, 0, 0])}.get(rank, (float("inf"), None))
    nodes = [10 * (rank + 1) + k for k in range(dim + 1)]
    a = reduce_enumeration(dist, cand[0], cand[1], nodes, dim)
    # case 2: a tie — the lowest rank's vector wins on every rank
    b = reduce_enumeration(dist, 1.0, [rank + 1] * dim, [1] * (dim + 1), dim)
    # case 3: nobody found anything
    c = reduce_enumeration(dist, float("inf"), None, [0] * (dim + 1), dim)
    q.put((rank, a, b, c))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_library_level_reductions_norm_vector_counts(world):
    """reduce_enumeration: norm MIN, the winner's vector to every rank, node counts SUM —
    identical results on all ranks, also with a tie and with no solution at all."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_reduce_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    dim = 6
    want_nodes = [sum(10 * (r + 1) + k for r in range(world)) for k in range(dim + 1)]
    for rank, a, b, c in res:
        assert a == (2.25, [0.0, 1.0, 1.0, -1.0, 0.0, 0.0], want_nodes)
        assert b == (1.0, [1.0] * dim, [world] * (dim + 1))
        assert c[0] == float("inf") and c[1] is None and c[2] == [0] * (dim + 1)


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    from fplll_amd.distributed import balance_plan, make_gather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = make_gather(dist, "cpu")
    # 1. the counts: 8 bytes per rank
    counts0 = [1000, 10, 0, 430][:world]
    blocks = g(int(counts0[rank]).to_bytes(8, "little"))
    counts = [int.from_bytes(b, "little") for b in blocks]
    assert counts == counts0
    # 2. the plan every rank derives, and the tasks moving by it: a "task" is 1040 bytes stamped (rank, index)
    moved, surplus, deficit, offset = balance_plan(counts)
    mine = [bytes([rank]) + i.to_bytes(4, "little") + bytes(1035) for i in range(counts[rank])]
    send = b"".join(mine[counts[rank] - surplus[rank]:])
    mine = mine[:counts[rank] - surplus[rank]]
    pool = b"".join(g(send))
    assert len(pool) == moved * 1040
    take = pool[offset[rank] * 1040:(offset[rank] + deficit[rank]) * 1040]
    mine += [take[i * 1040:(i + 1) * 1040] for i in range(deficit[rank])]
    # 3. ragged / empty blocks
    rag = g(bytes([rank]) * (rank * 3))
    assert [len(b) for b in rag] == [r * 3 for r in range(world)] and all(set(b) <= {r} for r, b in enumerate(rag))
    assert g(b"") == [b""] * world
    q.put((rank, [(t[0], int.from_bytes(t[1:5], "little")) for t in mine]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_work_movement_plan_and_gather(world):
    """The all-gather the work movement rides on (make_gather) and the plan rebalance_tasks derives from the
    counts (balance_plan, the Python restatement): after one round of movement the lists differ by at most one
    task, every task is held by exactly one rank, and nothing moves when the lists are level already."""
    import torch.multiprocessing as mp
    from fplll_amd.distributed import balance_plan
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    counts0 = [1000, 10, 0, 430][:world]
    sizes = [len(o[1]) for o in out]
    assert sum(sizes) == sum(counts0) and max(sizes) - min(sizes) <= 1
    everything = sorted(t for o in out for t in o[1])
    assert everything == sorted((r, i) for r in range(world) for i in range(counts0[r]))
    # level lists stay where they are; so does a list that is a sixteenth off
    assert balance_plan([500, 500, 501])[0] == 0
    assert balance_plan([1000, 1000, 900, 1000])[0] == 0
    moved, surplus, deficit, offset = balance_plan([40, 0, 0, 0])
    assert moved == 30 and surplus == [30, 0, 0, 0] and deficit == [0, 10, 10, 10] and offset == [0, 0, 10, 20]
    # blocks above 64 rows: a receiver without room for the ancestors' rows keeps everybody where they are
    assert balance_plan([40, 0, 0, 0], room=[100, 100, 9, 100])[0] == 0
    assert balance_plan([40, 0, 0, 0], room=[0, 10, 10, 10])[0] == 30
    # FPHIP_MOVE_FRACTION lowers the bar: a single surplus task moves
    assert balance_plan([1000, 1000, 900, 1000], fraction=10**9)[:2] == (75, [25, 25, 0, 25])


def _bench(args, env_extra, timeout=300):
    import json
    import subprocess
    import sys
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(C.ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=C.ROOT)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


@pytest.mark.parametrize("n", [2, 4])
def test_bench_py_spawns_its_own_ranks_when_launched_plainly(n):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset) must not run as ONE rank and print
    n_gpus = 1: it starts N ranks itself, and the line carries the rank count the process group reports
    (--rank-check: the launcher's half only — no GPU here; gloo)."""
    r, lines = _bench(["--gpus", str(n), "--rank-check"], {"FPHIP_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    assert len(lines) == 1, "rank 0 prints ONE line"
    j = lines[0]
    assert j["n_gpus"] == n and j["ranks"]["reported_by_process_group"] == n
    assert j["ranks"]["launcher"].startswith("bench.py")


def test_bench_py_refuses_a_rank_count_that_is_not_gpus():
    """A launcher that started another number of ranks than --gpus says (the driver's N = 8 command with WORLD_SIZE
    = 1, or the reverse): refuse instead of printing a line with the wrong n_gpus."""
    r, lines = _bench(["--gpus", "4", "--rank-check"], {"FPHIP_BENCH_BACKEND": "gloo", "WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 2 and not lines and "refusing" in r.stderr
    r, lines = _bench(["--gpus", "4"], {"FPHIP_BENCH_BACKEND": "gloo", "WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode == 2 and not lines and "refusing" in r.stderr


def test_bench_py_under_torchrun_keeps_working():
    """The driver's launch line (torch.distributed.run) is untouched by the self-spawn path."""
    import json
    import subprocess
    import sys
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, FPHIP_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(C.ROOT, "bench.py"),
                        "--gpus", "2", "--rank-check"], capture_output=True, text=True, timeout=300, env=env, cwd=C.ROOT)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks"]["launcher"] == "external"
