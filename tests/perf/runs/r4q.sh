#!/bin/bash
# round 4, call q: the library of the final commit (gso_host.hip gained fphip_gso_slide_reduction_blocks) on the
# strategy-BKZ / slide / SD / in-loop / GSO / LLL / enumeration-plugin tests once more
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4q; mkdir -p $O
( time timeout 420 python -m pytest tests/test_bkzs_gpu.py tests/test_zz_slide_gpu.py tests/test_zz_sd_bkz_gpu.py tests/test_gso_gpu.py tests/test_lll_gpu.py tests/test_bkz_gpu.py tests/test_hh_gpu.py -q -m gpu -k "not nested3" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=" $O/tests.log | tail -3
( time timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; echo "smoke rc=$?"
