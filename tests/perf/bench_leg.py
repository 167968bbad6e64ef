"""One leg of bench.py by name, stand-alone (for gpurun calls with a timeout per leg):
    python tests/perf/bench_leg.py <leg> [batch]      legs: bkz20_batch hlll_batch pruner lll_batch bkz40"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import fplll_amd  # noqa: E402

leg = sys.argv[1]
ctx = fplll_amd.Context(0)
fn = {"bkz20_batch": bench.bkz20_batch, "hlll_batch": bench.hlll_batch, "pruner": bench.pruner_leg,
      "lll_batch": bench.lll_batch, "bkz40": bench.bkz_strategies_batch}[leg]
t = time.time()
res = fn(ctx, int(sys.argv[2])) if len(sys.argv) > 2 else fn(ctx)
res["leg_wall_s"] = time.time() - t
print(leg, json.dumps(res), flush=True)
ctx.close()
