"""prune() of the 60-dim block of tests/golden/prune_q180_k60_b60_p05.json: the host loop and the volume kernel
with several values of FPHIP_PRUNER_MIN_DEVICE_STEPS (batches below that many polynomial steps stay on the host).
    python tests/perf/prune_thresholds.py [thresholds ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as C  # noqa: E402
from fplll_amd import pruner as P  # noqa: E402

with open(os.path.join(C.GOLDEN, "prune_q180_k60_b60_p05.json")) as fh:
    j = json.load(fh)
hx = lambda v: np.array([float.fromhex(x) for x in v])  # noqa: E731
a = (float.fromhex(j["radius"]), float.fromhex(j["preproc_cost"]), hx(j["gso_r"]), float.fromhex(j["target"]),
     j["metric"], j["flags"])
want = hx(j["coefficients"])


def best(fn, reps=5):
    fn()
    t = 1e9
    for _ in range(reps):
        t0 = time.time()
        r = fn()
        t = min(t, time.time() - t0)
    return 1e3 * t, r


ms, r = best(lambda: P.prune(*a))
print("host loop: %.2f ms, parity %s" % (ms, np.array_equal(r.coefficients, want)), flush=True)
for thr in [int(x) for x in sys.argv[1:]] or [16000, 40000, 100000, 250000]:
    os.environ["FPHIP_PRUNER_MIN_DEVICE_STEPS"] = str(thr)
    eng = P.Engine(0)
    ms, r = best(lambda: P.prune(*a, engine=eng))
    print("volume kernel above %7d steps: %.2f ms, parity %s, (device jobs, host jobs, launches) %s"
          % (thr, ms, np.array_equal(r.coefficients, want), eng.stats()), flush=True)
    eng.close()
