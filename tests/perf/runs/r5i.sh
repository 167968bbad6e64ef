#!/bin/bash
# round 5, call i: work movement between ranks (gloo world 2 / 4 on the one GPU)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests/test_enum_multirank_gpu.py -x -q -m gpu -s > $O/multirank.log 2>&1; echo "multirank rc=$?"; grep -v "^$" $O/multirank.log | tail -30 | cut -c1-300
