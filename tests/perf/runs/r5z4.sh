#!/bin/bash
# round 5, call z4: two-wave workgroups for big NQ = 1 batches of the strategy-BKZ kernel: tests + the bench leg
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5z4; mkdir -p $O
timeout 200 python tests/perf/bench_leg.py bkz40 > $O/bkz40.log 2>&1; echo "rc=$?"; tail -1 $O/bkz40.log | cut -c1-400
timeout 300 python -m pytest tests/test_bkzs_gpu.py tests/test_zz_sd_bkz_gpu.py -q -m gpu --durations=4 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -7 $O/tests.log | cut -c1-160
