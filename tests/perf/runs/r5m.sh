#!/bin/bash
# round 5, call m: the sweep kernel's 8-byte rows through the ring (K_GRAM8 / K_AXPY8): parity of every row-width
# path, roofline by width
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5m; mkdir -p $O
timeout 600 python -m pytest tests/test_gso_gpu.py -x -q -m gpu > $O/gso_tests.log 2>&1; echo "gso tests rc=$?"; tail -8 $O/gso_tests.log | cut -c1-250
for m in "3 1" "1 1" "0 1" "0 0"; do set -- $m; echo "NARROW=$1 WIDE_RING=$2"; FPHIP_GSO_NARROW=$1 FPHIP_GSO_WIDE_RING=$2 timeout 120 python tests/perf/gso_roof.py 8192 2>&1 | tail -1; done | tee $O/roof_by_width.log
