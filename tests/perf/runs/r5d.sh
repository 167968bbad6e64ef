#!/bin/bash
# round 5, call d: shader clock of the lone wave / of the full batch (s_memtime against s_memrealtime)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5d; mkdir -p $O
cp exp/libPROF.so fplll_amd/lib/libfplll_hip.so
timeout 60 python tests/perf/lll_prof.py 120 1 > $O/prof_1.log 2>&1; echo "rc=$?"; cat $O/prof_1.log
timeout 120 python tests/perf/lll_prof.py 120 2048 > $O/prof_2048.log 2>&1; echo "rc=$?"; head -3 $O/prof_2048.log
