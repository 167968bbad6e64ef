#!/bin/bash
# round 4, call i: Gram passes of the LLL / BKZ kernels on the float mirror of bf (Lattice::f32ok) — parity, then
# A/B of the batched legs (FPHIP_GSO_NARROW=0 switches the narrow paths off)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
( time timeout 500 python -m pytest tests/test_lll_gpu.py tests/test_bkz_gpu.py tests/test_bkzs_gpu.py tests/test_zz_sd_bkz_gpu.py tests/test_zz_slide_gpu.py tests/test_a_configs_at_size_gpu.py -q -m gpu -k "not test_00 and not config5 and not config3 and not nested3 and not inloop and not block_parallel" ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=" $O/tests.log | tail -3
for nar in 3 0; do
  for leg in "lll_batch 2048" "bkz40 1024" "bkz20_batch 256"; do
    FPHIP_GSO_NARROW=$nar timeout 200 python tests/perf/bench_leg.py $leg 2>&1 | grep -o '"\(lattices\|reductions\)_per_s": [0-9.]*\|"kernel_s": [0-9.]*\|parity[a-z_0-9]*": [a-z]*' | tr '\n' ' ' | sed "s/^/narrow=$nar $leg: /"; echo
  done
done | tee $O/legs.log
