#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
timeout 300 python tests/perf/dropin_u_diag.py > $O/diag.log 2>&1; echo "rc=$?"; cat $O/diag.log | cut -c1-400
timeout 300 python -m pytest tests/test_lll_gpu.py -q -m gpu -k "session" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log | cut -c1-220
