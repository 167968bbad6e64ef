#!/bin/bash
# round 5, call h: enumeration up to FPLLL_MAX_ENUM_DIM = 256 (four registers per lane in the top walk above 128 rows)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
timeout 600 python -m pytest tests/test_enum_gpu.py -x -q -m gpu -k "larger_than_64" > $O/enum_tests.log 2>&1; echo "enum tests rc=$?"; tail -25 $O/enum_tests.log | cut -c1-250
