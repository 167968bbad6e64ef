#!/bin/bash
# round 5, call l: the whole GPU suite + smoke() on the round's final library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5l; mkdir -p $O
timeout 1150 python -m pytest tests -q -m gpu --durations=25 > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
