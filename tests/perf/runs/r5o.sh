#!/bin/bash
# round 5, call o: the batch hand-off test of the suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5o; mkdir -p $O
timeout 300 python -m pytest tests/test_bkzs_gpu.py -x -q -m gpu -s -k "handoff_service" > $O/t.log 2>&1; echo "rc=$?"; grep -v "^$" $O/t.log | tail -8 | cut -c1-300
