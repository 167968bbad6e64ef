#!/bin/bash
mkdir -p gpurun_out/r6r
for i in 1 2 3 4 5; do
  FPHIP_SUBS_SPLIT=1 FPHIP_SUBS_DONATE=0 timeout 100 python -m pytest tests/test_enum_gpu.py -q -m gpu --timeout=90 -x -k "fixture_parity or fixed_bound or shrinking or pruned_fixed or many_solutions or edge or larger_than_64 or more_than_63 or (wide_blocks and 130)" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-110 | tail -2
done > gpurun_out/r6r/loop_prealloc.log 2>&1
cat gpurun_out/r6r/loop_prealloc.log
