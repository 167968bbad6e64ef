#!/bin/bash
# round 5, call e: the whole GPU suite on the block streams + out-of-line entry points (one process), smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5e; mkdir -p $O
( time timeout 1300 python -m pytest tests -x -q -m gpu --durations=14 -s ) > $O/suite_full.log 2>&1
echo "suite rc=$?" >> $O/suite_full.log
grep -v "^$" $O/suite_full.log | grep -E "passed|failed|error|rc=|real|hand-off|s call|tour" | cut -c1-300 | tail -40
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
