#!/bin/bash
# round 5, call z2: bkzs_kernel<1> capped at 256 registers (two waves per SIMD)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5z2; mkdir -p $O
for b in 1024 2048; do timeout 200 python tests/perf/bench_leg.py bkz40 $b > $O/bkz40_$b.log 2>&1; echo "rc=$?"; tail -1 $O/bkz40_$b.log | cut -c100-330; done
timeout 300 python -m pytest tests/test_bkzs_gpu.py -q -m gpu -x --durations=4 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -7 $O/tests.log | cut -c1-160
