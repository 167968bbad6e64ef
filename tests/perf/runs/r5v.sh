#!/bin/bash
# round 5, call v: enum_phase_kernel without scratch memory — enumeration parity tests + the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5v; mkdir -p $O
timeout 500 python -m pytest tests/test_enum_gpu.py tests/test_a_configs_at_size_gpu.py -q -m gpu -x --durations=5 -k "not hlll" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -10 $O/tests.log | cut -c1-200
timeout 600 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-700
