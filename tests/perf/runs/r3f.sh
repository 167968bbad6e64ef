set -x
O=gpurun_out/r3f; mkdir -p $O
( time timeout 900 python -m pytest tests/test_zz_slide_gpu.py -x -q -m gpu -s ) > $O/tests_slide.log 2>&1
( time timeout 900 python -m pytest tests/test_zz_sd_bkz_gpu.py tests/test_bkzs_gpu.py -x -q -m gpu ) > $O/tests_sd_bkzs.log 2>&1
( time timeout 900 python -m pytest tests/test_dropin_gso_gpu.py -x -q -m gpu -k "not config2" ) > $O/tests_dropin.log 2>&1
( time timeout 900 python -m pytest tests/test_enum_multirank_gpu.py tests/test_enum_gpu.py -x -q -m gpu ) > $O/tests_enum.log 2>&1
( time FPHIP_BKZ_HANDOFF_NODES=10000 timeout 900 python tests/perf/c3_handoff.py ) > $O/c3_handoff_10k.log 2>&1
( time FPHIP_BKZ_HANDOFF_NODES=3000 timeout 900 python tests/perf/c3_handoff.py ) > $O/c3_handoff_3k.log 2>&1
