"""Diagnostic: the transformation matrix through the dropin (resident / stateless / host object)."""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest as C
DRV = os.path.join(ROOT, "oracle", "_ref", "dropin_driver")

def run(args, env):
    r = subprocess.run([DRV] + args, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    if r.returncode != 0:
        print("rc", r.returncode, r.stderr[-500:]); return None
    return json.loads(r.stdout)

def basis_file(b):
    path = tempfile.mktemp(suffix=".txt", dir="/tmp")
    with open(path, "w") as fh:
        fh.write("[" + "\n".join("[" + " ".join(str(int(x)) for x in row) + "]" for row in b) + "]\n")
    return path

for what, fx, args in (("lll", "lll_q72.json", ["lll", None, "hip"]), ("bkz", "bkz_q60_b16.json", ["bkz", None, "16", "hip"])):
    f = (C.load_lll_fixture if what == "lll" else C.load_bkz_fixture)(os.path.join(C.GOLDEN, fx))
    path = basis_file(f["b_in"]); d = f["d"]
    res = {}
    for tag, a, env in (("resident", "hip", {"DROPIN_U": "1"}), ("stateless", "hip", {"DROPIN_U": "1", "FPLLL_HIP_RESIDENT": "0"}),
                        ("host", "cpu", {"DROPIN_U": "1"})):
        aa = [x if x is not None else path for x in args]; aa[-1 if what == "lll" else 3] = a
        res[tag] = run(aa, env)
    os.unlink(path)
    uc = np.array(res["host"]["u_out"], dtype=np.int64).reshape(d, d)
    for tag in ("resident", "stateless"):
        j = res[tag]
        if j is None: continue
        u = np.array(j["u_out"], dtype=np.int64).reshape(d, d)
        b = np.array(j["b_out"], dtype=np.int64).reshape(d, -1)
        bad = [i for i in range(d) if not np.array_equal(u[i], uc[i])]
        print(what, tag, "b ok", np.array_equal(b, f["b_out"]), "u == host's", not bad, "bad rows", bad[:12], len(bad),
              "u b_in == b_out", np.array_equal(u.astype(object).dot(f["b_in"].astype(object)), b.astype(object)),
              "calls", j["device_calls"], "starts", j.get("session_starts"), "dirty", j.get("dirty_rows"))
