#!/bin/bash
# round 4, call e: fused AXPY+Gram pass of the sweep kernel (parity, roofline A/B, occupancy sweep), and the new
# bench legs one by one under a timeout (call d's bench did not finish)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
( time timeout 300 python -m pytest tests/test_gso_gpu.py tests/test_lll_gpu.py tests/test_a_configs_at_size_gpu.py -q -m gpu -k "not test_00 and not config2 and not config5 and not config3" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=" $O/tests.log | tail -3
for m in 3 2; do FPHIP_GSO_NARROW=$m timeout 200 python tests/perf/gso_roof.py 8192 2>&1 | tail -1 | sed "s/^/narrow=$m /"; done | tee $O/roof.log
for bpc in 1 2 3; do FPHIP_GSO_BLOCKS_PER_CU=$bpc timeout 200 python tests/perf/gso_roof.py 8192 2>&1 | tail -1 | sed "s/^/blocks_per_cu=$bpc /"; done | tee -a $O/roof.log
for leg in "hlll_batch 256" "bkz20_batch 256" ; do
  ( time timeout 200 python tests/perf/bench_leg.py $leg ) > $O/leg_$(echo $leg | tr ' ' '_').log 2>&1; echo "$leg rc=$?"; tail -4 $O/leg_$(echo $leg | tr ' ' '_').log | cut -c1-500
done
for w in 2 8; do
  ( time FPHIP_BKZ_PRUNE_WORKERS=$w timeout 200 python tests/perf/bench_leg.py pruner ) > $O/leg_pruner_w$w.log 2>&1; echo "pruner workers=$w rc=$?"; tail -4 $O/leg_pruner_w$w.log | cut -c1-900
done
