set -x
O=gpurun_out/r3k; mkdir -p $O
( time timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-gso --no-tour ) > $O/bench_quick.log 2> $O/bench_quick.err
( time timeout 600 python -m pytest tests/test_enum_gpu.py tests/test_enum_multirank_gpu.py -x -q -m gpu ) > $O/tests.log 2>&1
tail -3 $O/tests.log
cut -c1-400 $O/bench_quick.log | tail -2
