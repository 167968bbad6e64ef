#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5y; mkdir -p $O
for i in 1 2 3; do FPHIP_NOTES=1 timeout 300 python -m pytest tests/test_enum_multirank_gpu.py -q -m gpu -s -k "above_64" > $O/tests$i.log 2>&1; echo "rc=$?"; grep -i "block of\|passed\|failed" $O/tests$i.log | cut -c1-200; done
