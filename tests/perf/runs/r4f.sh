#!/bin/bash
# round 4, call f: helper streams in their own priority class (the 8-worker in-loop run of call e deadlocked behind
# the schedule kernel's hardware queue), sweep occupancy A/B repeated
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4f; mkdir -p $O
for w in 4 8; do
  ( time FPHIP_BKZ_PRUNE_WORKERS=$w timeout 150 python tests/perf/bench_leg.py pruner ) > $O/leg_pruner_w$w.log 2>&1; echo "pruner workers=$w rc=$?"; grep -o '"bkz40_inloop.*' $O/leg_pruner_w$w.log | cut -c1-400
done
for rep in 1 2 3; do for bpc in 3 4; do FPHIP_GSO_BLOCKS_PER_CU=$bpc timeout 200 python tests/perf/gso_roof.py 8192 2>&1 | tail -1 | sed "s/^/rep $rep blocks_per_cu=$bpc /"; done; done | tee $O/roof.log
( time timeout 600 python -m pytest tests/test_bkzs_gpu.py tests/test_zz_slide_gpu.py tests/test_zzz_long_runs_gpu.py -q -m gpu -k "inloop or block_parallel or handoff" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=" $O/tests.log | tail -3
