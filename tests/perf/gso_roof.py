"""Roofline run of the batched size-reduction sweep (fplll_amd.gso.bench_roofline), stand-alone:
    python tests/perf/gso_roof.py [batch ...]            -> one line per batch
    python tests/perf/gso_roof.py --once <batch>         -> ONE launch (for rocprofv3 --pmc passes)
FPHIP_GSO_SWEEP=1 selects the first-generation kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fplll_amd  # noqa: E402
from fplll_amd import gso as G  # noqa: E402

args = sys.argv[1:]
once = "--once" in args
args = [a for a in args if a != "--once"]
ctx = fplll_amd.Context(0)
for B in [int(a) for a in args] or [8192]:
    r = G.bench_roofline(ctx, batch=B, reps=1 if once else 2)
    print("B=%d kernel_ms=%.2f achieved=%.1f GB/s frac=%.3f (with confirming pass: %.3f)"
          % (B, r["kernel_ms"], r["achieved"], r["frac"], r["with_confirming_pass"]["frac"]), flush=True)
    if os.environ.get("FPHIP_ROOF_JSON"):
        print(json.dumps(r), flush=True)
ctx.close()
