#!/bin/bash
# round 5, call b: block streams + Gram entries one by one + diagonal in the Gram pass; per-phase timers (libPROF);
# the first generation (libOLD) on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5b; mkdir -p $O
timeout 240 python -m pytest tests/test_lll_gpu.py tests/test_bkz_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 120 python tests/perf/lll_bench.py 120 2048 1 > $O/lll_new_2048.log 2>&1; echo "rc=$?"; tail -2 $O/lll_new_2048.log
timeout 60 python tests/perf/lll_bench.py 120 1 1 > $O/lll_new_1.log 2>&1; echo "rc=$?"; tail -2 $O/lll_new_1.log
cp fplll_amd/lib/libfplll_hip.so /tmp/new.so
cp exp/libPROF.so fplll_amd/lib/libfplll_hip.so
timeout 60 python tests/perf/lll_prof.py 120 1 > $O/prof_1.log 2>&1; echo "rc=$?"; cat $O/prof_1.log
timeout 120 python tests/perf/lll_prof.py 120 2048 > $O/prof_2048.log 2>&1; echo "rc=$?"; cat $O/prof_2048.log
cp exp/libOLD.so fplll_amd/lib/libfplll_hip.so
timeout 120 python tests/perf/lll_bench.py 120 2048 0 > $O/lll_old_2048.log 2>&1; echo "rc=$?"; tail -1 $O/lll_old_2048.log
timeout 60 python tests/perf/lll_bench.py 120 1 0 > $O/lll_old_1.log 2>&1; echo "rc=$?"; tail -1 $O/lll_old_1.log
cp /tmp/new.so fplll_amd/lib/libfplll_hip.so
