set -x
O=gpurun_out/r3g; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/suite.log 2>&1
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err
