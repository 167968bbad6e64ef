set -x
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
( time timeout 600 python -m pytest tests/test_bkzs_gpu.py tests/test_zz_sd_bkz_gpu.py tests/test_enum_gpu.py -x -q -m gpu ) > $O/tests.log 2>&1
# baseline tour at several MIN_NODES, with stats
for mn in 200000 50000 20000 5000; do
  FPLLL_HIP_STATS=1 FPLLL_HIP_MIN_NODES=$mn timeout 300 python tests/perf/bkz_tour.py 60 fplll_amd/lib/libfplll_hip_extenum.so > $O/tour_$mn.log 2>&1
done
timeout 300 python tests/perf/bkz_tour.py 60 none > $O/tour_cpu.log 2>&1
# pruner regime launch anatomy
FPHIP_DEBUG=1 timeout 300 python bench.py --regime pruner --no-cpu --no-gso --no-tour --no-pmc --steps 3 --warmup 1 > $O/pruner_dbg.log 2>&1
timeout 300 python bench.py --regime pruner --no-cpu --no-gso --no-tour --no-pmc --steps 6 --warmup 1 > $O/pruner.log 2>&1
