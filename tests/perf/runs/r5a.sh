#!/bin/bash
# round 5, call a: the block streams of lll_stream.h — LLL parity tests, then batched / single LLL against the
# first generation's library (exp/libOLD.so) on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
timeout 240 python -m pytest tests/test_lll_gpu.py -x -q -m gpu > $O/lll_tests.log 2>&1; echo "lll tests rc=$?"; tail -5 $O/lll_tests.log
timeout 120 python tests/perf/lll_bench.py 120 2048 1 > $O/lll_new_2048.log 2>&1; echo "rc=$?"; tail -3 $O/lll_new_2048.log
timeout 60 python tests/perf/lll_bench.py 120 1 1 > $O/lll_new_1.log 2>&1; echo "rc=$?"; tail -3 $O/lll_new_1.log
timeout 200 python -m pytest tests/test_bkz_gpu.py -x -q -m gpu > $O/bkz_tests.log 2>&1; echo "bkz tests rc=$?"; tail -5 $O/bkz_tests.log
cp fplll_amd/lib/libfplll_hip.so /tmp/new.so; cp exp/libOLD.so fplll_amd/lib/libfplll_hip.so
timeout 120 python tests/perf/lll_bench.py 120 2048 0 > $O/lll_old_2048.log 2>&1; echo "rc=$?"; tail -2 $O/lll_old_2048.log
timeout 60 python tests/perf/lll_bench.py 120 1 0 > $O/lll_old_1.log 2>&1; echo "rc=$?"; tail -2 $O/lll_old_1.log
cp /tmp/new.so fplll_amd/lib/libfplll_hip.so
