set -x
O=gpurun_out/r3d; mkdir -p $O
( time timeout 900 python -m pytest tests/test_enum_gpu.py tests/test_enum_multirank_gpu.py tests/test_reference_kats.py -x -q -m gpu ) > $O/tests.log 2>&1
( time timeout 900 python -m pytest tests/test_dropin_gso_gpu.py -x -q -m gpu -k "not config2" ) > $O/tests_dropin.log 2>&1
( time timeout 600 python -m pytest tests/test_a_configs_at_size_gpu.py -x -q -m gpu -k "config3 and (pruner or linear)" ) > $O/tests_c3.log 2>&1
P="python bench.py --regime pruner --no-cpu --no-gso --no-tour --no-pmc"
FPHIP_DEBUG=1 timeout 300 $P --steps 3 --warmup 1 > $O/pruner_dbg.log 2>&1
timeout 300 $P --steps 6 --warmup 1 > $O/pruner.log 2>&1
FPHIP_BFS=0 FPHIP_DEBUG=1 timeout 300 $P --steps 3 --warmup 1 > $O/pruner_nobfs_dbg.log 2>&1
FPHIP_BFS=0 timeout 300 $P --steps 6 --warmup 1 > $O/pruner_nobfs.log 2>&1
for hv in 64 1024 4096; do FPHIP_BFS_HEAVY=$hv timeout 300 $P --steps 6 --warmup 1 > $O/pruner_hv$hv.log 2>&1; done
FPHIP_BFS_WG_PER_CU=2 timeout 300 $P --steps 6 --warmup 1 > $O/pruner_wg2.log 2>&1
FPHIP_BFS_WG_PER_CU=1 timeout 300 $P --steps 6 --warmup 1 > $O/pruner_wg1.log 2>&1
FPHIP_BFS_SINGLE_MAX=2048 timeout 300 $P --steps 6 --warmup 1 > $O/pruner_single2048.log 2>&1
M="python bench.py --no-cpu --no-gso --no-tour --no-pmc --steps 3 --warmup 1"
FPHIP_DEBUG=1 timeout 300 $M > $O/main_dbg.log 2>&1
for mn in 200000 50000 20000 5000; do
  FPLLL_HIP_STATS=1 FPLLL_HIP_MIN_NODES=$mn timeout 300 python tests/perf/bkz_tour.py 60 fplll_amd/lib/libfplll_hip_extenum.so > $O/tour_$mn.log 2>&1
done
