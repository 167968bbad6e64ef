#!/bin/bash
# round 5, call p: BKZ-20 batch at 2048 (two waves per SIMD), config-3 hand-off tours at batch 64
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5p; mkdir -p $O
timeout 200 python tests/perf/bench_leg.py bkz20_batch 2048 > $O/bkz20_2048.log 2>&1; echo "rc=$?"; tail -1 $O/bkz20_2048.log | cut -c1-500
timeout 500 python tests/perf/c3_handoff_batch.py 64 4 > $O/c3_b64.log 2>&1; echo "rc=$?"; tail -1 $O/c3_b64.log | cut -c1-600
