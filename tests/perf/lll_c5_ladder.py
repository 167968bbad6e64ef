"""BASELINE config 5's lattice (256-dim NTRU-like, `latticegen n 128 12 b`) through the LLL-side precision
ladder on the device (fphip_gso_lll_ladder): LLLReduction<long,double> stops with RED_BABAI_FAILURE on it
(in the reference: "infinite loop in babai"), the double-double stage reduces it.  Prints times, statuses
and the reference's reducedness predicate on the output."""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import conftest as C  # noqa: E402
import fplll_amd  # noqa: E402
from fplll_amd.gso import MatGSOBatch  # noqa: E402
import test_dd_gpu as T  # noqa: E402

f = C.load_hlll_fixture(os.path.join(C.GOLDEN, "c5_hlll_n256_double.json.gz"))
b = f["b_in"]
ctx = fplll_amd.Context(0)
g = MatGSOBatch(ctx, 1, 256, 256)
mode = sys.argv[1] if len(sys.argv) > 1 else "ladder"
g.set_basis(np.stack([b]))
t = time.time()
if mode == "ladder":
    st, info, stage = g.lll_ladder()
else:
    st, info = g.lll_ex(106)
    stage = [106]
wall = time.time() - t
out = g.get_basis(0, 1)[0]
res = {"mode": mode, "status": int(st[0]), "stage": int(stage[0]), "info": [int(v) for v in info[0]],
       "wall_s": wall, "kernel_ms_last_stage": g.last_kernel_ms}
if int(st[0]) == 1:
    res["stat"] = T._basisstat(out)
    res["in_lattice"] = bool(T._rows_in_qary_lattice(b, out))
print(json.dumps(res))
