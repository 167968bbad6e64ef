#!/bin/bash
# round 5, call j: the deep LDS-DMA ring for launches of a few lattices (one wave per workgroup, 64 KiB)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
timeout 300 python -m pytest tests/test_lll_gpu.py -x -q -m gpu > $O/lll_tests.log 2>&1; echo "lll tests rc=$?"; tail -3 $O/lll_tests.log
timeout 60 python tests/perf/lll_bench.py 120 1 1 > $O/lll_deep_1.log 2>&1; echo "rc=$?"; tail -2 $O/lll_deep_1.log
FPHIP_LLL_DEEP_RING=0 timeout 60 python tests/perf/lll_bench.py 120 1 0 > $O/lll_shallow_1.log 2>&1; echo "rc=$?"; tail -1 $O/lll_shallow_1.log
timeout 60 python tests/perf/lll_bench.py 180 1 0 > $O/lll_deep_1_d180.log 2>&1; echo "rc=$?"; tail -1 $O/lll_deep_1_d180.log
FPHIP_LLL_DEEP_RING=0 timeout 60 python tests/perf/lll_bench.py 180 1 0 > $O/lll_shallow_1_d180.log 2>&1; echo "rc=$?"; tail -1 $O/lll_shallow_1_d180.log
cp exp/libPROF.so fplll_amd/lib/libfplll_hip.so
timeout 60 python tests/perf/lll_prof.py 120 1 > $O/prof_1.log 2>&1; echo "rc=$?"; cat $O/prof_1.log
