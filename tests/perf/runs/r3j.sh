set -x
O=gpurun_out/r3j; mkdir -p $O
( time FPHIP_DEBUG=1 timeout 600 python tests/perf/c3_handoff.py ) > $O/c3_handoff.log 2>&1
( time FPHIP_BKZ_HANDOFF_NODES=800 timeout 600 python tests/perf/c3_handoff.py ) > $O/c3_handoff_800.log 2>&1
( time timeout 600 python -m pytest tests/test_bkzs_gpu.py tests/test_zz_sd_bkz_gpu.py tests/test_zz_slide_gpu.py tests/test_dropin_gso_gpu.py -x -q -m gpu -k "not config2" ) > $O/tests.log 2>&1
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err
