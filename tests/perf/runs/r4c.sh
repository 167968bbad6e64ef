#!/bin/bash
# round 4, call c: sweep kernel with double rows / anchored transpose / bk from the 16-bit row (parity + roofline,
# A/B against the mirror widths), in-loop pruning with the worker pool and the work threshold
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gso_gpu.py tests/test_lll_gpu.py tests/test_pruner_gpu.py tests/test_a_configs_at_size_gpu.py tests/test_bkz_gpu.py -q -s -m gpu -k "not test_00 and not config2 and not config5 and not config3" --durations=8 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|prune\(\) of|rc=" $O/tests.log | tail -5
for i in 1 2; do timeout 300 python tests/perf/gso_roof.py 8192 2>&1 | tail -1; done | tee $O/roof.log
FPHIP_GSO_NARROW=1 timeout 300 python tests/perf/gso_roof.py 8192 2>&1 | tail -1 | tee -a $O/roof.log
FPHIP_GSO_NARROW=0 timeout 300 python tests/perf/gso_roof.py 8192 2>&1 | tail -1 | tee -a $O/roof.log
( time timeout 600 python -m pytest tests/test_bkzs_gpu.py -q -s -m gpu -k "inloop" ) > $O/inloop.log 2>&1
echo "inloop rc=$?" >> $O/inloop.log
grep -E "passed|failed|rc=" $O/inloop.log | tail -3
grep -E "prune calls" $O/inloop.log | cut -c1-330 | tail -8
