#!/bin/bash
# round 5, call c: register streams (raw buffer loads, 16 rows in flight, no LDS), out-of-line LLL entry points in
# the BKZ kernels; parity tests, batched / single LLL, per-phase timers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5c; mkdir -p $O
timeout 300 python -m pytest tests/test_lll_gpu.py tests/test_bkz_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 120 python tests/perf/lll_bench.py 120 2048 1 > $O/lll_new_2048.log 2>&1; echo "rc=$?"; tail -2 $O/lll_new_2048.log
timeout 60 python tests/perf/lll_bench.py 120 1 1 > $O/lll_new_1.log 2>&1; echo "rc=$?"; tail -2 $O/lll_new_1.log
timeout 400 python -m pytest tests/test_bkzs_gpu.py -x -q -m gpu > $O/bkzs_tests.log 2>&1; echo "bkzs tests rc=$?"; tail -3 $O/bkzs_tests.log
cp fplll_amd/lib/libfplll_hip.so /tmp/new.so
cp exp/libPROF.so fplll_amd/lib/libfplll_hip.so
timeout 60 python tests/perf/lll_prof.py 120 1 > $O/prof_1.log 2>&1; echo "rc=$?"; cat $O/prof_1.log
timeout 120 python tests/perf/lll_prof.py 120 2048 > $O/prof_2048.log 2>&1; echo "rc=$?"; cat $O/prof_2048.log
cp /tmp/new.so fplll_amd/lib/libfplll_hip.so
