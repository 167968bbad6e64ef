"""Multi-rank balance on ONE GPU (gloo): node shares and kernel ms per rank for the headline blocks, with the
breadth-first stage sharded (FPHIP_BFS_SHARD=1) and replicated (=0).  usage: shard_balance2.py world"""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def worker(rank, world, port, name, q):
    import torch.distributed as dist
    import conftest as C
    import fplll_amd
    from fplll_amd.distributed import make_exchange, make_gather
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = C.load_fixture(os.path.join(C.GOLDEN, name + ".json"))
    ctx = fplll_amd.Context(0)
    out = []
    for rep in range(2):
        ev = FastEvaluator(1, 0)
        res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev, shard_index=rank, shard_count=world,
                              exchange=make_exchange(dist, "cpu"), exchange_chunks=4, gather=make_gather(dist, "cpu"))
        out.append((int(res.total_nodes), float(res.stats.kernel_ms), float(res.stats.wall_ms), int(res.stats.moved_tasks),
                    min([s[0] for s in ev.solutions] or [float("inf")])))
    q.put((rank, out))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    import torch.multiprocessing as mp
    world = int(sys.argv[1])
    for name in ("c3_b60_k0_linear30", "c3_b60_k1_pruner"):
        for mode in ("1", "0"):
            os.environ["FPHIP_BFS_SHARD"] = mode
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            c = mp.get_context("spawn")
            q = c.Queue()
            ps = [c.Process(target=worker, args=(r, world, port, name, q)) for r in range(world)]
            [p.start() for p in ps]
            res = sorted(q.get(timeout=600) for _ in range(world))
            [p.join(60) for p in ps]
            last = [r[1][-1] for r in res]
            tot = sum(x[0] for x in last)
            print(json.dumps({"block": name, "world": world, "bfs_sharded": mode == "1", "total_nodes": tot,
                              "node_shares": [round(x[0] / max(1, tot), 3) for x in last],
                              "kernel_ms": [round(x[1], 2) for x in last], "wall_ms": [round(x[2], 2) for x in last],
                              "moved": [x[3] for x in last], "best": [x[4] for x in last]}), flush=True)
