"""Batched LLL throughput: B independent 120-dim q-ary lattices (the C2 lattice family of
BASELINE.json: `latticegen q 120 60 20 p`), LLLReduction::lll on each.  Prints lattices/s and, when
oracle/_ref/ref_driver is present, the real reference's single-core time on the same inputs (and
checks that the output bases are identical)."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fplll_amd
from fplll_amd.gso import MatGSOBatch

def qary(rng, d, k, q):
    b = np.zeros((d, d), dtype=np.int64)
    b[:k, :k] = np.eye(k, dtype=np.int64)
    b[:k, k:] = rng.integers(0, q, size=(k, d - k))
    b[k:, k:] = q * np.eye(d - k, dtype=np.int64)
    return b

def main():
    d = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    ncheck = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    q = 1048583  # a 21-bit prime (gen_qary_prime draws a 20-bit random and takes the next prime)
    rng = np.random.default_rng(0)
    bs = np.stack([qary(rng, d, d // 2, q) for _ in range(B)])
    ctx = fplll_amd.Context(0)
    g = MatGSOBatch(ctx, B, d, d)
    g.set_basis(bs)
    t = time.perf_counter()
    st, info = g.lll()
    wall = time.perf_counter() - t
    ms = g.last_kernel_ms
    assert np.all(st == 1), np.unique(st, return_counts=True)
    print("d=%d B=%d: lll kernel %.1f ms (wall %.1f ms) -> %.1f lattices/s; swaps mean %.0f, iterations mean %.0f"
          % (d, B, ms, wall * 1e3, B / (ms * 1e-3), info[:, 1].mean(), info[:, 3].mean()), flush=True)
    out = g.get_basis(0, B)
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if os.path.exists(drv) and ncheck > 0:
        secs = []
        for L in range(ncheck):
            with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
                f.write("[" + "\n".join("[" + " ".join(str(int(x)) for x in row) + "]" for row in bs[L]) + "]\n")
                path = f.name
            r = subprocess.run([drv, "lllfix", "f:" + path, "0", "0", "0", "0", "0", "0", "-1", "0", "0"],
                               capture_output=True, text=True, timeout=600)
            os.unlink(path)
            j = json.loads(r.stdout)
            ref = np.array(j["b_out"], dtype=np.int64).reshape(d, d)
            assert np.array_equal(ref, out[L]), "lattice %d differs from the reference" % L
            assert j["n_swaps"] == info[L][1]
            secs.append(j["ref_seconds"])
        print("reference (1 core): %.3f s per lattice -> %.2f lattices/s; outputs identical on %d checked; speedup %.0fx"
              % (np.mean(secs), 1 / np.mean(secs), ncheck, (B / (ms * 1e-3)) * np.mean(secs)), flush=True)
    g.close(); ctx.close()

main()
