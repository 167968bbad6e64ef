#!/bin/bash
# round 4, call p: the block-parallel slide tour behind the C ABI (fphip_gso_slide_reduction_blocks)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4p; mkdir -p $O
( time timeout 300 python -m pytest tests/test_zz_slide_gpu.py -q -m gpu -k "c_abi or contexts" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|Error|assert" $O/tests.log | tail -8
