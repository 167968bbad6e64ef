#!/bin/bash
# round 4, call k: the final commit under the driver's conditions — the whole GPU suite in one process, smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4k; mkdir -p $O
( time timeout 1300 python -m pytest tests -x -q -m gpu --durations=12 ) > $O/suite.log 2>&1
echo "suite rc=$?" >> $O/suite.log
tail -5 $O/suite.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( timeout 300 python -m pytest tests/test_zzz_long_runs_gpu.py -q -s -m gpu -k handoff 2>&1 | grep -E "hand-off:|passed|failed" | cut -c1-400 ) | tee $O/handoff.log
