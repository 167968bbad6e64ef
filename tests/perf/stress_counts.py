"""Stress: repeat a fixed-radius enumeration (exact counts known) in 2 concurrent processes on one
GPU, unsharded, to expose timing-dependent races."""
import os, sys, time
import multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def worker(rank, reps, name, q, mode='plain'):
    import conftest as C
    import fplll_amd
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    f = C.load_fixture(os.path.join(C.GOLDEN, name + ".json"))
    ctx = fplll_amd.Context(0)
    bad = []
    for i in range(reps):
        import numpy as np
        if mode == "chunks":
            ev = FastEvaluator(f["max_sols"], f["strategy"])
            res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev, exchange_chunks=3)
            n = [int(v) for v in res.nodes]
        elif mode == "shards":  # both shards computed by THIS process, one after the other
            tot = None
            for sh in range(2):
                ev = FastEvaluator(f["max_sols"], f["strategy"])
                res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev,
                                      shard_index=sh, shard_count=2, exchange_chunks=3)
                tot = res.nodes.copy() if tot is None else tot + res.nodes
            n = [int(v) for v in tot]
        else:
            ev = FastEvaluator(f["max_sols"], f["strategy"])
            res = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev)
            n = [int(v) for v in res.nodes]
        if n != f["nodes"]:
            diff = [(k, a - b) for k, (a, b) in enumerate(zip(n, f["nodes"])) if a != b]
            bad.append((i, diff[:6], len(diff), res.stats.phases))
    q.put((rank, len(bad), bad[:3]))
    ctx.close()

if __name__ == "__main__":
    nproc = int(sys.argv[1]); reps = int(sys.argv[2]); name = sys.argv[3] if len(sys.argv) > 3 else "enum_d48_lin30_fixed"
    mode = sys.argv[4] if len(sys.argv) > 4 else "plain"
    c = mp.get_context("spawn"); q = c.Queue()
    ps = [c.Process(target=worker, args=(r, reps, name, q, mode)) for r in range(nproc)]
    [p.start() for p in ps]
    for _ in ps:
        print(q.get(timeout=600), flush=True)
    [p.join() for p in ps]
