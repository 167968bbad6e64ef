#!/bin/bash
mkdir -p gpurun_out/r6r
for i in 1 2 3 4 5 6 7 8; do
  timeout 300 python -m pytest tests/test_enum_gpu.py -q -m gpu --timeout=200 -x -k "fixture_parity or fixed_bound or shrinking or pruned_fixed or many_solutions or edge or larger_than_64 or more_than_63 or wide_blocks" > gpurun_out/r6r/fix_$i.log 2>&1
  grep -E "passed|failed|AssertionError: per" gpurun_out/r6r/fix_$i.log | cut -c1-160 | tail -2
done
timeout 200 python tests/perf/wide_subs_debug.py 3 2>&1 | tail -6
