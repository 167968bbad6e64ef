#!/bin/bash
# round 4, call t: FPHIP_PRUNER_MIN_DEVICE_STEPS 16000 (default) against 100000 on the in-loop leg, and the
# pruner / in-loop tests with 100000
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4t; mkdir -p $O
for thr in 16000 100000; do
  FPHIP_PRUNER_MIN_DEVICE_STEPS=$thr timeout 40 python tests/perf/bench_leg.py pruner > $O/leg_pruner_$thr.log 2>&1; echo "leg $thr rc=$?"; tail -c 520 $O/leg_pruner_$thr.log | head -c 400; echo
done
( time FPHIP_PRUNER_MIN_DEVICE_STEPS=100000 timeout 55 python -m pytest tests/test_pruner_gpu.py tests/test_bkzs_gpu.py -q -m gpu -k "pruner or inloop" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|Error|assert" $O/tests.log | tail -8
