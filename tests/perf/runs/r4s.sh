#!/bin/bash
# round 4, call s: the host volume loop eight jobs wide (pruner_volume.hip) — the pruner / in-loop tests on the
# rebuilt library, prune() timings with several device thresholds, the pruner leg of bench.py
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4s; mkdir -p $O
( time timeout 60 python -m pytest tests/test_pruner_gpu.py tests/test_bkzs_gpu.py -q -m gpu -k "pruner or inloop or volumes or engine" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=" $O/tests.log | tail -3
timeout 40 python tests/perf/prune_thresholds.py > $O/thresholds.log 2>&1; echo "thresholds rc=$?"; cat $O/thresholds.log
timeout 50 python tests/perf/bench_leg.py pruner > $O/leg_pruner.log 2>&1; echo "leg rc=$?"; tail -c 900 $O/leg_pruner.log
