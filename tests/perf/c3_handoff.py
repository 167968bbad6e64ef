"""BASELINE config 3's BKZ-60 tour on the device in hand-off mode (FPHIP_BKZ_HANDOFF), alone on the GPU:
wall time, nodes, and the reference's reducedness predicate on the output (ref_driver basisstat)."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_a_configs_at_size_gpu as A  # noqa: E402
out = {}
A._run_config3_tour_handoff(out)
print(json.dumps(out, default=str))
