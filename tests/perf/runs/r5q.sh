#!/bin/bash
# round 5, call q: LLL_EARLY_RED on the device — the LLL test file, the dropin variants, a quick batched-LLL rate check
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5q; mkdir -p $O
timeout 400 python -m pytest tests/test_lll_gpu.py tests/test_dropin_gso_gpu.py -q -m gpu -x --durations=8 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -12 $O/tests.log | cut -c1-200
timeout 200 python tests/perf/bench_leg.py lll_batch > $O/lll_batch.log 2>&1; echo "rc=$?"; tail -1 $O/lll_batch.log | cut -c1-400
