"""One BKZ-60 tour of the 180-dim q-ary lattice with the REFERENCE's bkz_reduction (oracle/_ref):
(a) fplll's internal enumerator, (b) our HIP enumerator installed through set_external_enumerator.
SURVEY.md §8(d) metric (ii).  Needs oracle/_ref (travels with the tree) and a GPU for (b)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
basis = os.path.join(ROOT, "tests", "golden", "basis_q180_seed0_lll_bkz20.txt")
strat = os.path.join(ROOT, "tests", "golden", "strategies_q180_b60.json")
so = os.path.join(ROOT, "fplll_amd", "lib", "libfplll_hip_extenum.so")
beta = sys.argv[1] if len(sys.argv) > 1 else "60"
which = sys.argv[2:] or ["none", so]
out = {}
for plug in which:
    env = dict(os.environ)
    env.setdefault("FPLLL_HIP_MIN_NODES", "200000")
    t = time.time()
    r = subprocess.run([drv, "bkztour", basis, strat, beta, plug], capture_output=True, text=True, env=env, timeout=3000)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    j = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
    j["wall_incl_load"] = round(time.time() - t, 3)
    out["plugin" if plug == so else plug] = j
    print(json.dumps(j), flush=True)
if "none" in out and "plugin" in out and "tour_seconds" in out["none"] and "tour_seconds" in out["plugin"]:
    print(json.dumps({"tour_speedup": out["none"]["tour_seconds"] / out["plugin"]["tour_seconds"],
                      "same_basis": out["none"]["basis_fnv"] == out["plugin"]["basis_fnv"],
                      "r00_ratio": out["plugin"]["r00"] / out["none"]["r00"]}))
