#!/bin/bash
# round 4, call w (the last GPU seconds): task buffers allocated by the first fphip_enum_run — smoke() and the
# enumeration tests that use every one of those buffers (blocks above 64, shards, overflow, edge cases)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4w; mkdir -p $O
timeout 12 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; grep -c "^smoke:" $O/smoke.log
timeout 21 python -m pytest tests/test_enum_gpu.py -q -m gpu -x -k "sharded_counts or task_buffer_overflow or edge_cases or larger_than_64 or multi_device" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
