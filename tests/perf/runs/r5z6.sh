#!/bin/bash
# round 5, call z6: the default bench line at the round's last commit
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5z6; mkdir -p $O
( time timeout 280 python bench.py > $O/bench.log 2> $O/bench.err ) 2> $O/time.log; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-200; grep real $O/time.log
