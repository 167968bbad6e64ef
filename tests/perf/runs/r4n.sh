#!/bin/bash
# round 4, call n: pruner_volume.hip was recompiled (host-side change only): the pruner's device tests and the
# in-loop tests once more on the rebuilt library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O
( time timeout 400 python -m pytest tests/test_pruner_gpu.py tests/test_bkzs_gpu.py -q -m gpu -k "pruner or prune or volumes or inloop" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=" $O/tests.log | tail -3
