#!/bin/bash
mkdir -p gpurun_out/r6r
for i in 1 2 3 4 5 6 7 8; do
  FPHIP_SUBS_SPLIT=1 FPHIP_SUBS_DONATE=0 timeout 300 python -m pytest tests/test_enum_gpu.py -q -m gpu --timeout=200 -x -k "fixture_parity or fixed_bound or shrinking or pruned_fixed or many_solutions or edge or larger_than_64 or more_than_63 or (wide_blocks and 130) or subsolutions" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-110 | tail -3
done > gpurun_out/r6r/loop_b.log 2>&1
cat gpurun_out/r6r/loop_b.log
