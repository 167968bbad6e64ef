#!/bin/bash
# round 4, call a: the regression fix under the driver's conditions, minus the two long runs:
# new MatGSOBatch member tests, the RCCL world-size-1 test, the NQ=4 tests, smoke, and the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gso_gpu.py tests/test_enum_multirank_gpu.py tests/test_a_configs_at_size_gpu.py tests/test_dd_gpu.py tests/test_hh_gpu.py tests/test_lll_gpu.py -q -m gpu -k "not test_00 and not config2 and not config5 and not bench_py" -x --durations=15 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log
( time timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -5 $O/tests.log; tail -3 $O/smoke.log; tail -c 3000 $O/bench.log; tail -5 $O/bench.err
