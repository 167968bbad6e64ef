#!/bin/bash
# round 5, call f: the resident LLL session (fphip_gso_session_lll) and MatGSOHip on it: config 2 through the
# reference's unmodified bkz()
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5f; mkdir -p $O
timeout 300 python -m pytest tests/test_lll_gpu.py -x -q -m gpu > $O/lll_tests.log 2>&1; echo "lll tests rc=$?"; tail -15 $O/lll_tests.log | cut -c1-200
( time timeout 600 python -m pytest tests/test_dropin_gso_gpu.py -x -q -m gpu -s --durations=8 ) > $O/dropin.log 2>&1; echo "dropin rc=$?"; grep -v "^$" $O/dropin.log | tail -25 | cut -c1-300
