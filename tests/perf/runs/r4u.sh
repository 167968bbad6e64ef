#!/bin/bash
# round 4, call u (the round's last GPU seconds): prune() timings with the FMA quotient in the host volume loop
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4u; mkdir -p $O
timeout 35 python tests/perf/prune_thresholds.py 16000 100000 > $O/thresholds.log 2>&1; echo "thresholds rc=$?"; cat $O/thresholds.log
FPHIP_PRUNER_HOST_MODE=1 timeout 12 python tests/perf/prune_thresholds.py 16000 > $O/thresholds_divider.log 2>&1; tail -2 $O/thresholds_divider.log
