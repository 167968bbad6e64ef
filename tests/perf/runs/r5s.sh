#!/bin/bash
# round 5, call s: u through sessions and the dropin (the reference's bkz() with a transformation matrix)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5s; mkdir -p $O
timeout 500 python -m pytest tests/test_lll_gpu.py tests/test_dropin_gso_gpu.py -q -m gpu --durations=6 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -40 $O/tests.log | cut -c1-220
