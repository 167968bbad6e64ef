#!/bin/bash
# round 5, call z5: the 80-dim HLLL leg at 4096 lattices (16 waves per CU by registers)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5z5; mkdir -p $O
timeout 150 python - > $O/hlll_4096.log 2>&1 <<'PY'
import sys, json
sys.path.insert(0, "tests")
import bench, fplll_amd
ctx = fplll_amd.Context(0)
print("hlll_batch", json.dumps(bench.hlll_batch(ctx, 4096, 80)))
ctx.close()
PY
echo "rc=$?"; tail -1 $O/hlll_4096.log | cut -c1-420
