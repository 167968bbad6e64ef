set -x
O=gpurun_out/r3r; mkdir -p $O
( time timeout 120 python tests/perf/lll_bench.py 120 1024 1 ) > $O/lll_bench.log 2>&1; tail -3 $O/lll_bench.log | cut -c1-300
( time timeout 120 python tests/perf/bkz_bench.py ) > $O/bkz_bench.log 2>&1; tail -3 $O/bkz_bench.log | cut -c1-300
( time timeout 120 python tests/perf/hlll_bench.py ) > $O/hlll_bench.log 2>&1; tail -3 $O/hlll_bench.log | cut -c1-300
