import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import conftest as C
import fplll_amd
from fplll_amd.householder import MatHouseholderBatch
from fplll_amd.gso import _unreduced_copy
import test_gso_gpu as T
ctx = fplll_amd.Context(0)
full = T._load_c3_basis()
b = _unreduced_copy(full, 2, 9)
d = 180
t = time.time(); C.oracle_hh_update_all(b, True); cpu = time.time() - t
for B in (1024, 4096):
    h = MatHouseholderBatch(ctx, B, d, d, row_expo=True)
    h.set_basis(b); h.broadcast_basis(0)
    h.update_R(); ms = h.last_kernel_ms
    h.update_R(); ms = min(ms, h.last_kernel_ms)
    # algorithmic bytes: row i reads reflectors V[0..i) tails: sum_j 8(n-j) + writes V_i, R_i (16 n)
    alg = sum(sum(8 * (d - j) for j in range(i)) + 16 * d for i in range(d)) * B
    print("B=%d exact kernel %.2f ms -> %.0f lattices/s, %.1f GB/s algorithmic; C oracle 1 core %.2f ms/lattice (%.0f x)"
          % (B, ms, B / ms * 1e3, alg / ms / 1e6, cpu * 1e3, cpu * 1e3 / (ms / B)), flush=True)
    # the opt-in blocked (MFMA compact-WY) mode: 4/3 d^3 flops of the QR per lattice
    h.update_R(blocked=True); mb = h.last_kernel_ms
    h.update_R(blocked=True); mb = min(mb, h.last_kernel_ms)
    flops = B * (4.0 / 3.0) * d ** 3
    print("B=%d blocked MFMA kernel %.2f ms -> %.0f lattices/s, %.2f TFLOP/s (f64), %.1fx the exact kernel"
          % (B, mb, B / mb * 1e3, flops / mb / 1e9, ms / mb), flush=True)
    h.close()
