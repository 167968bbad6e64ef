#!/bin/bash
# round 4, call b: the rebuilt pruner on the device, in-loop pruning of the BKZ service, plus the tests
# call a did not reach
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4b; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_pruner_gpu.py tests/test_bkzs_gpu.py tests/test_gso_gpu.py tests/test_enum_multirank_gpu.py tests/test_a_configs_at_size_gpu.py tests/test_dd_gpu.py tests/test_hh_gpu.py tests/test_lll_gpu.py tests/test_zz_slide_gpu.py tests/test_zz_sd_bkz_gpu.py -q -s -m gpu -k "not test_00 and not config2 and not config5 and not bench_py" --durations=25 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|prune\(\) of|rc=" $O/tests.log | tail -8
grep -E "prune calls" $O/tests.log | cut -c1-400 | tail -8
