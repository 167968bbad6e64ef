#!/bin/bash
# round 5, call z3: bkzs_kernel<1> at 256 registers AND smaller workgroups (LDS packing): residency 6-7 waves per CU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5z3; mkdir -p $O
for cfg in "2 1536" "1 1792" "2 3072"; do set -- $cfg; FPHIP_GSO_WAVES_PER_BLOCK=$1 timeout 200 python tests/perf/bench_leg.py bkz40 $2 > $O/bkz40_w$1_$2.log 2>&1; echo "wpb=$1 batch=$2 rc=$?"; tail -1 $O/bkz40_w$1_$2.log | cut -c150-290; done
