#!/bin/bash
# round 4, call o: gso_host.hip recompiled (error-path clean-up only): in-loop, slide-pass and strategy tests again
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4o; mkdir -p $O
( time timeout 400 python -m pytest tests/test_bkzs_gpu.py tests/test_zz_slide_gpu.py tests/test_gso_gpu.py tests/test_lll_gpu.py -q -m gpu -k "not nested3" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=" $O/tests.log | tail -3
