set -x
O=gpurun_out/r3m; mkdir -p $O
( time timeout 600 python -m pytest tests/test_enum_gpu.py tests/test_enum_multirank_gpu.py -x -q -m gpu ) > $O/tests_enum.log 2>&1
tail -2 $O/tests_enum.log
( time timeout 300 python tests/perf/bkzs_bench.py 1024 ) > $O/bkzs_bench.log 2>&1
tail -3 $O/bkzs_bench.log
( time timeout 600 python -m pytest tests/test_bkzs_gpu.py tests/test_zz_sd_bkz_gpu.py tests/test_zz_slide_gpu.py -x -q -m gpu ) > $O/tests_bkzs.log 2>&1
tail -2 $O/tests_bkzs.log
