"""Two ranks (two processes, gloo collectives) sharing ONE GPU run the sharded enumeration exactly
as bench.py --gpus 2 does: content-sorted partition of the subtree tasks + bound/active all-reduce at
every chunk and round boundary.  With a radius that never shrinks the ranks' per-level node counts
must add up to the reference's counts — disjoint and complete — and the collective must terminate
(same number of exchange calls on every rank)."""
import os
import socket

import numpy as np
import pytest

import conftest as C

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fixture, q):
    import torch.distributed as dist
    import fplll_amd
    from fplll_amd.distributed import make_exchange
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = C.load_fixture(fixture)
    ctx = fplll_amd.Context(0)
    calls = [0]
    ex0 = make_exchange(dist, "cpu")

    def ex(b, a):
        calls[0] += 1
        return ex0(b, a)

    ev = FastEvaluator(f["max_sols"], f["strategy"])
    res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev,
                          shard_index=rank, shard_count=world, exchange=ex, exchange_chunks=3)
    q.put((rank, [int(v) for v in res.nodes], calls[0], len(ev.solutions),
           (res.stats.phases, res.stats.final_tasks, res.stats.final_root_level, res.stats.overflowed)))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,share", [("enum_d48_lin30_fixed", 0.2),
                                        # blocks above 64 levels: the top walk (levels >= 64) is
                                        # replicated on every rank and counted on shard 0 only
                                        ("enum_d80_lin70_fixed", 0.05), ("enum_d96_lin90_fixed", 0.0)])
def test_two_ranks_partition_the_tree(name, share):
    import torch.multiprocessing as mp
    fixture = os.path.join(C.GOLDEN, name + ".json")
    f = C.load_fixture(fixture)
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    ps = [mpctx.Process(target=_worker, args=(r, 2, port, fixture, q)) for r in range(2)]
    for p in ps:
        p.start()
    out = [q.get(timeout=300) for _ in range(2)]
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    out.sort()
    tot = np.array(out[0][1]) + np.array(out[1][1])
    diff = [(k, int(a) - b) for k, (a, b) in enumerate(zip(tot, f["nodes"])) if int(a) != b]
    assert not diff, (str(diff), out[0][2], out[1][2], out[0][4], out[1][4])
    assert min(sum(out[0][1]), sum(out[1][1])) > share * f["total_nodes"]  # both ranks did real work
    assert out[0][2] == out[1][2] >= 3  # identical collective call counts


def _worker_lib(rank, world, port, fixture, q):
    import torch.distributed as dist
    import fplll_amd
    from fplll_amd.distributed import enumerate_block_sharded
    from fplll_amd.enumeration import FastEvaluator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = C.load_fixture(fixture)
    ctx = fplll_amd.Context(0)
    ev = FastEvaluator(f["max_sols"], f["strategy"])
    gd, gc, gn, res = enumerate_block_sharded(ctx, dist, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev,
                                              device="cpu", exchange_chunks=3)
    q.put((rank, gd, gc, gn, int(res.total_nodes)))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("enum_d48_lin30_fixed", 3), ("enum_d36_lin18_best1", 2),
                                        ("enum_d48_lin30_best1", 4)])
def test_sharded_call_with_library_level_reductions(name, world):
    """enumerate_block_sharded = what bench.py --gpus N times: the sharded walk, then norm MIN -> the
    winner's vector to every rank -> node counts SUM.  Every rank ends with the SAME
    result; at a fixed radius the summed per-level counts are the reference's, with a shrinking one the
    final norm is."""
    import torch.multiprocessing as mp
    fixture = os.path.join(C.GOLDEN, name + ".json")
    f = C.load_fixture(fixture)
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    ps = [mpctx.Process(target=_worker_lib, args=(r, world, port, fixture, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    for r in range(1, world):
        assert out[r][1:4] == out[0][1:4], "ranks disagree on the reduced result"
    gd, gc, gn = out[0][1:4]
    assert sum(o[4] for o in out) == sum(gn)
    ref_best = min([s[0] for s in f["sol_log"]] or [float("inf")])
    if "fixed" in name:
        assert gn == [int(v) for v in f["nodes"]], "summed per-level counts differ from the reference's"
        assert gd <= ref_best  # (the evaluator keeps every solution here: the shortest is among them)
    elif ref_best == float("inf"):  # (the reference finds nothing inside this radius)
        assert gd == ref_best and gc is None
    else:
        assert gd == ref_best and gc is not None and len(gc) == f["d"]


@pytest.mark.parametrize("world,launcher", [(2, "torchrun"), (4, "plain")])
def test_bench_py_multi_rank_protocol_on_one_device(world, launcher):
    """bench.py exactly as the driver launches it for N GPUs (torch.distributed.run, one rank per
    process) — with the collectives on gloo and every rank on the ONE GPU of this box
    (FPHIP_BENCH_BACKEND=gloo, FPHIP_BENCH_ONE_DEVICE=1): the sharded call, the bound exchange, the
    library-level reductions and the max-over-ranks timing.  Every timed step must end on the
    reference's final norm, like the N = 1 line; nodes are the whole job's.  "plain": `python bench.py --gpus N`
    with no launcher (WORLD_SIZE unset) starts its N ranks itself and reports the rank count of the process group —
    a plain launch must never come back as n_gpus = 1."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, FPHIP_BENCH_BACKEND="gloo", FPHIP_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    tail = [os.path.join(C.ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
            "--no-cpu", "--no-gso", "--no-tour", "--no-pmc"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + tail
    else:
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=C.ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE json line"
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["steps"] == 3 and j["scaling"] == "strong"
    assert j["ranks"]["reported_by_process_group"] == world
    assert j["ranks"]["launcher"] == ("external" if launcher == "torchrun" else "bench.py (self-spawned)")
    assert j["parity"]["final_norm_equal_to_reference"] == 3, j["parity"]
    assert j["nodes"] > 3 * 10**9 and j["value"] > 0


def _worker_rccl(port, fixture, q):
    import torch
    import torch.distributed as dist
    import fplll_amd
    from fplll_amd.distributed import enumerate_block_sharded, gather_status
    from fplll_amd.enumeration import FastEvaluator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    f = C.load_fixture(fixture)
    ctx = fplll_amd.Context(0)
    ev = FastEvaluator(f["max_sols"], f["strategy"])
    gd, gc, gn, res = enumerate_block_sharded(ctx, dist, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev,
                                              device="cuda:0", exchange_chunks=3)
    st = gather_status(dist, [1, 1, 1], 3, 0, 1, device="cuda:0")
    q.put((gd, gc, gn, int(res.total_nodes), st, dist.get_backend()))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["enum_d48_lin30_fixed", "enum_d36_lin18_best1"])
def test_sharded_call_on_rccl_world_size_one(name):
    """The RCCL code path itself, executed on hardware: backend "nccl" (= RCCL on ROCm),
    world size 1 — process-group initialisation on the GPU, the 16-byte all_reduce(MIN) of the bound
    exchange on a DEVICE tensor at every chunk / round boundary, and the three result reductions
    (all_gather of (norm, vector), all_reduce(SUM) of the counts) — around the same sharded call the
    gloo tests run.  Counts / final norm are the reference's."""
    import torch.multiprocessing as mp
    fixture = os.path.join(C.GOLDEN, name + ".json")
    f = C.load_fixture(fixture)
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    p = mpctx.Process(target=_worker_rccl, args=(_free_port(), fixture, q))
    p.start()
    gd, gc, gn, total, st, backend = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    assert backend == "nccl" and st == [1, 1, 1]
    assert total == sum(gn)
    ref_best = min([s[0] for s in f["sol_log"]] or [float("inf")])
    if "fixed" in name:
        assert gn == [int(v) for v in f["nodes"]]
        assert gd <= ref_best
    else:
        assert gd == ref_best


def _load_case(fixture):
    """A golden fixture, or "wide:<d>:<seed>" = conftest.wide_block_with_candidates (no reference run exists for
    blocks above 128 rows: the C oracle is the checker there)."""
    if not fixture.startswith("wide:"):
        return C.load_fixture(fixture)
    _, d, seed = fixture.split(":")
    mut, rdiag, maxdist = C.wide_block_with_candidates(int(d), int(seed))
    return {"mut": mut, "rdiag": rdiag, "pruning": None, "maxdist": maxdist, "d": int(d), "name": fixture}


def _worker_move(rank, world, port, fixture, move, fixed, q):
    import torch.distributed as dist
    import fplll_amd
    from fplll_amd.distributed import make_exchange, make_gather
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = _load_case(fixture)
    ctx = fplll_amd.Context(0)
    ev = FastEvaluator(10**9 if fixed else 1, 0)
    res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev,
                          shard_index=rank, shard_count=world, exchange=make_exchange(dist, "cpu"), exchange_chunks=2,
                          gather=make_gather(dist, "cpu") if move else None)
    best = min([s[0] for s in ev.solutions] or [float("inf")])
    sols = sorted((float(s[0]), tuple(float(v) for v in s[1])) for s in ev.solutions) if fixed else []
    q.put((rank, [int(v) for v in res.nodes], int(res.stats.moved_tasks), float(res.stats.kernel_ms), best, sols))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


def _run_move(name, world, move, fixed):
    import torch.multiprocessing as mp
    fixture = name if name.startswith("wide:") else os.path.join(C.GOLDEN, name + ".json")
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    ps = [mpctx.Process(target=_worker_move, args=(r, world, port, fixture, move, fixed, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=600) for _ in range(world))
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    return _load_case(fixture), out


@pytest.mark.parametrize("world", [2, 4])
def test_donated_subtrees_move_between_ranks(world):
    """fphip_enum_opts::gather: at every round boundary of the walk the ranks level their lists of donated
    subtrees (surplus tasks travel as 1040-byte records through an all-gather; enumlib's shared subtree counter,
    enum-parallel/enumeration.h:412-505, between processes).  At a radius that never shrinks the per-level counts
    of the ranks still add up to the reference's — every subtree walked exactly once, wherever — tasks did move,
    and no rank is left with a sliver of the work."""
    f, out = _run_move("enum_d48_lin30_fixed", world, True, True)
    tot = np.sum([np.array(o[1]) for o in out], axis=0)
    assert [int(v) for v in tot] == [int(v) for v in f["nodes"]]
    shares = [sum(o[1]) / f["total_nodes"] for o in out]
    C.note(lambda: ("work movement, %d ranks on one GPU, %s: node shares %s, tasks moved per rank %s, kernel ms %s"
                    % (world, f["name"], ["%.3f" % s for s in shares], [o[2] for o in out], ["%.1f" % o[3] for o in out]),))
    assert min(shares) > 0.5 / world
    # the same call without movement: the same counts (the A/B of the protocol)
    f2, out2 = _run_move("enum_d48_lin30_fixed", world, False, True)
    tot2 = np.sum([np.array(o[1]) for o in out2], axis=0)
    assert [int(v) for v in tot2] == [int(v) for v in f["nodes"]] and all(o[2] == 0 for o in out2)


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_breadth_first_stage_partitions_the_tree(world, monkeypatch):
    """FPHIP_BFS_SHARD=1 (opt-in): the breadth-first stage itself is sharded — its thin top (one workgroup) stays
    replicated, but a final task it emits is kept by ONE rank (the rank its coefficient prefix hashes to), and from
    the first level launched over the chip every rank expands only its share of the frontier; nothing is dealt
    afterwards, the lists are levelled by the work movement.  At a radius that never shrinks the per-level counts of
    the ranks add up to the reference's: every node expanded once, every subtree walked once.  (The default keeps
    the stage replicated and deals the final tasks in snake order: better balanced on small trees — the A/B is in
    DESIGN.md section 3, profiles/r06_bfs_shard_balance.log.)"""
    monkeypatch.setenv("FPHIP_BFS_SHARD", "1")
    for name in ("enum_d48_lin30_fixed", "enum_d40_lin20_fixed"):
        f, out = _run_move(name, world, True, True)
        tot = np.sum([np.array(o[1]) for o in out], axis=0)
        assert [int(v) for v in tot] == [int(v) for v in f["nodes"]], name
        got = sorted(x for o in out for x in o[5])
        want = sorted((float(s[0]), tuple(float(v) for v in s[1])) for s in f["sol_log"])
        assert got == want
        C.note(lambda: ("sharded breadth-first stage, %d ranks, %s: node shares %s"
                        % (world, name, ["%.3f" % (sum(o[1]) / f["total_nodes"]) for o in out]),))


@pytest.mark.parametrize("name", ["enum_d96_lin90_fixed", "enum_d80_lin70_fixed"])
def test_donated_subtrees_of_a_block_above_64_rows_move_with_their_ancestors(name, monkeypatch):
    """A task of a block above 64 rows points into its rank's table of level-64 ancestors (the coefficients of the
    levels >= 64), filled by a top walk that runs in an order of its own on every rank: the record of a moving task
    carries that row and the receiver appends it to its table.  Few, large first tasks and the smallest donation
    budget make the tasks shed work in every round, and the bar for a transfer is lowered to a single surplus task, so
    that lists exist to be levelled whatever the timing; at a radius that never shrinks the per-level counts of the
    ranks add up to the reference's and the candidates (whose coefficients above level 64 come out of the moved rows)
    are the reference's."""
    monkeypatch.setenv("FPHIP_BUDGET", "256")
    monkeypatch.setenv("FPHIP_BFS_TASKS", "96")
    monkeypatch.setenv("FPHIP_MOVE_FRACTION", "1000000000")
    f, out = _run_move(name, 2, True, True)
    tot = np.sum([np.array(o[1]) for o in out], axis=0)
    assert [int(v) for v in tot] == [int(v) for v in f["nodes"]]
    ref_best = min([s[0] for s in f["sol_log"]] or [float("inf")])
    assert min(o[4] for o in out) == ref_best
    # every candidate of the reference, coefficients included, exactly once over the ranks
    got = sorted(x for o in out for x in o[5])
    want = sorted((float(s[0]), tuple(float(v) for v in s[1])) for s in f["sol_log"])
    assert got == want
    C.note(lambda: ("work movement, block of %d rows, 2 ranks: node shares %s, tasks moved per rank %s"
                    % (f["d"], ["%.3f" % (sum(o[1]) / f["total_nodes"]) for o in out], [o[2] for o in out]),))
    assert sum(o[2] for o in out) > 0, "no task moved: the case does not exercise the records of wide blocks"


@pytest.mark.parametrize("d,seed", [(160, 41), (200, 42)])
def test_blocks_above_128_rows_shard_and_move_with_their_ancestor_rows(d, seed, monkeypatch):
    """Two ranks on a block above 128 rows with candidates under several level-64 ancestors: the sharding keys
    (task_key_kernel), the moved records (task_pack / task_unpack, 128 or 192 doubles of ancestor coefficients) and
    the solution reports all read the ancestor table with the same row stride, so the ranks agree on the partition —
    per-level counts add up to the C oracle's, every candidate of the oracle exactly once, coefficients included."""
    from fplll_amd.enumeration import FastEvaluator
    monkeypatch.setenv("FPHIP_BUDGET", "256")
    monkeypatch.setenv("FPHIP_MOVE_FRACTION", "1000000000")
    f, out = _run_move("wide:%d:%d" % (d, seed), 2, True, True)
    ev_o = FastEvaluator(10**9, 0)
    nodes_o, _ = C.oracle_enumerate(f["mut"], f["rdiag"], None, f["maxdist"], ev_o)
    tot = np.sum([np.array(o[1]) for o in out], axis=0)
    # (the top walk is replicated and counted by shard 0 only)
    assert [int(v) for v in tot] == [int(v) for v in nodes_o]
    got = sorted(x for o in out for x in o[5])
    want = sorted((float(s[0]), tuple(float(v) for v in s[1])) for s in ev_o.solutions)
    assert len(want) >= 30 and got == want
    C.note(lambda: ("block of %d rows, 2 ranks: node shares %s, tasks moved per rank %s"
                    % (d, ["%.3f" % (sum(o[1]) / max(1, sum(int(v) for v in nodes_o))) for o in out], [o[2] for o in out]),))


def test_work_movement_with_a_shrinking_radius_and_on_a_pruner_regime_block():
    """With BEST-1 semantics the final norm over the ranks is the reference's whether or not tasks move; on a
    small (pruner-regime) block of config 3 every rank still ends with the shortest vector."""
    for name in ("enum_d48_lin30_best1", "c3_b60_k2_pruner"):
        f, out = _run_move(name, 2, True, False)
        ref_best = min([s[0] for s in f["sol_log"]] or [float("inf")])
        assert min(o[4] for o in out) <= ref_best  # (never longer; a parallel walk may end shorter)
        if name.startswith("enum_"):
            assert min(o[4] for o in out) == ref_best
        C.note(lambda: ("work movement, 2 ranks, %s: node shares %s, moved %s"
                        % (name, ["%.3f" % (sum(o[1]) / max(1, sum(sum(x[1]) for x in out))) for o in out],
                           [o[2] for o in out]),))
