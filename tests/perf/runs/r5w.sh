#!/bin/bash
# round 5, call w: work movement for blocks above 64 rows
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5w; mkdir -p $O
FPHIP_NOTES=1 timeout 500 python -m pytest tests/test_enum_multirank_gpu.py -q -m gpu -x --durations=5 -s > $O/tests.log 2>&1; echo "tests rc=$?"; grep -i "work movement\|passed\|failed\|Error" $O/tests.log | cut -c1-260 | tail -15
