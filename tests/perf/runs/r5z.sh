#!/bin/bash
# round 5, call z: BKZ-40 with strategies at 2048 / 4096 lattices
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5z; mkdir -p $O
for b in 2048 4096; do timeout 200 python tests/perf/bench_leg.py bkz40 $b > $O/bkz40_$b.log 2>&1; echo "rc=$?"; tail -1 $O/bkz40_$b.log | cut -c1-420; done
