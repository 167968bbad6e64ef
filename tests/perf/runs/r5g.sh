#!/bin/bash
# round 5, call g: LLL_SIEGEL on the device (fixtures of the driven reference), sessions with flags, the drop-in
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
timeout 300 python -m pytest tests/test_lll_gpu.py -x -q -m gpu > $O/lll_tests.log 2>&1; echo "lll tests rc=$?"; tail -12 $O/lll_tests.log | cut -c1-220
timeout 300 python -m pytest tests/test_dropin_gso_gpu.py -x -q -m gpu -k "lll or stateless" > $O/dropin.log 2>&1; echo "dropin rc=$?"; tail -8 $O/dropin.log | cut -c1-220
