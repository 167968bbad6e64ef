#!/bin/bash
# round 5, call n: the hand-off service shared by a batch of config-3 tours (a pool of enumeration contexts)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5n; mkdir -p $O
timeout 200 python tests/perf/c3_handoff_batch.py 1 1 > $O/b1.log 2>&1; echo "rc=$?"; tail -1 $O/b1.log | cut -c1-600
timeout 400 python tests/perf/c3_handoff_batch.py 16 3 > $O/b16.log 2>&1; echo "rc=$?"; tail -1 $O/b16.log | cut -c1-600
FPHIP_BKZ_HANDOFF_WORKERS=1 timeout 400 python tests/perf/c3_handoff_batch.py 16 1 > $O/b16_w1.log 2>&1; echo "rc=$?"; tail -1 $O/b16_w1.log | cut -c1-600
