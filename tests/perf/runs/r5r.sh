#!/bin/bash
# round 5, call r: the transformation matrix u on the device (LLL)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5r; mkdir -p $O
timeout 400 python -m pytest tests/test_lll_gpu.py -q -m gpu -x --durations=5 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/tests.log | cut -c1-220
