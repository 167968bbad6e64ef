"""Batched BKZ throughput on the C2 lattice family of BASELINE.json (`latticegen q 120 60 20 p`,
BKZ-20, BKZ_DEFAULT, no strategies): device LLL then device BKZ on B independent lattices; checks
lattice 0.. against the real reference (oracle/_ref/ref_driver bkzfix on the same input) when it is
present, and prints its single-core time."""
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
    beta = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    ncheck = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    q = 1048583
    rng = np.random.default_rng(0)
    bs = np.stack([qary(rng, d, d // 2, q) for _ in range(B)])
    ctx = fplll_amd.Context(0)
    g = MatGSOBatch(ctx, B, d, d)
    g.set_basis(bs)
    st, _ = g.lll()
    lll_ms = g.last_kernel_ms
    assert np.all(st == 1)
    st, info = g.bkz(beta)
    ms = g.last_kernel_ms
    assert np.all(st == 1), np.unique(st, return_counts=True)
    nodes = (info[:, 1].astype(np.int64) & 0xffffffff) | ((info[:, 2].astype(np.int64) & 0xffffffff) << 32)
    print("d=%d beta=%d B=%d: lll %.1f ms, bkz kernel %.1f ms -> %.2f BKZ/s; tours mean %.1f, nodes mean %.3g, enum calls mean %.0f"
          % (d, beta, B, lll_ms, ms, B / (ms * 1e-3), info[:, 0].mean(), nodes.mean(), info[:, 3].mean()), flush=True)
    out = g.get_basis(0, B)
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if os.path.exists(drv) and ncheck > 0:
        secs = []
        for L in range(ncheck):
            with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
                f.write("[" + "\n".join("[" + " ".join(str(int(x)) for x in row) + "]" for row in bs[L]) + "]\n")
                path = f.name
            r = subprocess.run([drv, "bkzfix", "f:" + path, "0", "0", "0", "0", str(beta), "0"],
                               capture_output=True, text=True, timeout=1200)
            os.unlink(path)
            j = json.loads(r.stdout)
            ref = np.array(j["b_out"], dtype=np.int64).reshape(d, d)
            assert np.array_equal(ref, out[L]), "lattice %d differs from the reference" % L
            assert j["nodes"] == int(nodes[L]), (j["nodes"], int(nodes[L]))
            secs.append(j["ref_seconds"])
        print("reference (1 core): %.3f s per BKZ -> %.3f BKZ/s; output basis and node count identical on %d checked; speedup %.1fx"
              % (np.mean(secs), 1 / np.mean(secs), ncheck, (B / (ms * 1e-3)) * np.mean(secs)), flush=True)
    g.close(); ctx.close()

main()
