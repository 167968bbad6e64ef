set -x
O=gpurun_out/r3h; mkdir -p $O
( time timeout 600 python -m pytest tests/test_dd_gpu.py -x -q -m gpu -s -k "lll_in_double" ) > $O/tests_lllx.log 2>&1
( time timeout 700 python tests/perf/lll_c5_ladder.py ladder ) > $O/c5_ladder.log 2>&1
( time timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu --no-tour --no-pmc ) > $O/bench_legs.log 2> $O/bench_legs.err
( time timeout 300 python tests/perf/lll_bench.py 120 1024 0 ) > $O/lll_base.log 2>&1
cp fplll_amd/lib/libfplll_hip.so /tmp/libbase.so
cp exp/libUNI.so fplll_amd/lib/libfplll_hip.so
( time timeout 300 python tests/perf/lll_bench.py 120 1024 0 ) > $O/lll_uni.log 2>&1
( time timeout 300 python -m pytest tests/test_lll_gpu.py tests/test_bkz_gpu.py tests/test_hlll_gpu.py -x -q -m gpu ) > $O/tests_uni.log 2>&1
cp /tmp/libbase.so fplll_amd/lib/libfplll_hip.so
