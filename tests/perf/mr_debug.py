import os, sys, socket
import multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def worker(rank, world, port, reps, q):
    import numpy as np
    import torch.distributed as dist
    import conftest as C
    import fplll_amd
    from fplll_amd.distributed import make_exchange
    from fplll_amd.enumeration import FastEvaluator, enumerate_block
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    f = C.load_fixture(os.path.join(C.GOLDEN, "enum_d48_lin30_fixed.json"))
    ctx = fplll_amd.Context(0)
    def run(sh, ex):
        ev = FastEvaluator(f["max_sols"], f["strategy"])
        r = enumerate_block(ctx, f["mut"], f["rdiag"], f["pruning"], f["maxdist"], ev, shard_index=sh, shard_count=world,
                            exchange=ex, exchange_chunks=3)
        return np.array([int(v) for v in r.nodes])
    ex = make_exchange(dist, "cpu")
    mode = os.environ.get("MR_MODE", "ex")
    swap = int(os.environ.get("MR_SWAP", "0"))
    myshard = (rank + swap) % world
    if mode == "ex":
        cold = run(myshard, ex)                           # COLD first call, concurrent, with collectives
    elif mode == "noex":
        cold = run(rank, None)                            # cold, concurrent, no collectives
    else:                                                 # "solo": only rank 0 runs cold, rank 1 waits
        cold = run(rank, None) if rank == 0 else None
        dist.barrier()
        if rank == 1:
            cold = run(rank, None)
    dist.barrier()
    exp = [run(s, None) for s in range(world)]          # warm, sequential, no collectives
    ok_seq = [int(v) for v in sum(exp)] == f["nodes"]
    d = cold.astype(np.int64) - exp[myshard if mode == 'ex' else rank].astype(np.int64)
    q.put((rank, "shard=%d seq_ok=%s" % (myshard if mode == 'ex' else rank, ok_seq), int(d.sum()), [(k, int(v)) for k, v in enumerate(d) if v][:4]))
    ctx.close(); dist.destroy_process_group()

if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    c = mp.get_context("spawn"); q = c.Queue()
    ps = [c.Process(target=worker, args=(r, 2, port, int(sys.argv[1]), q)) for r in range(2)]
    [p.start() for p in ps]
    for _ in ps: print(q.get(timeout=900), flush=True)
    [p.join() for p in ps]
