set -x
O=gpurun_out/r3e; mkdir -p $O
( time timeout 900 python -m pytest tests/test_dropin_gso_gpu.py -x -q -m gpu -k "not config2" ) > $O/tests_dropin.log 2>&1
( time FPHIP_DEBUG=1 timeout 900 python tests/perf/c3_handoff.py ) > $O/c3_handoff.log 2>&1
( time FPHIP_BKZ_HANDOFF_NODES=30000 timeout 900 python tests/perf/c3_handoff.py ) > $O/c3_handoff_30k.log 2>&1
P="python bench.py --regime pruner --no-cpu --no-gso --no-tour --no-pmc --steps 6 --warmup 1"
for cfg in "1024 1" "1024 2" "2048 1" "512 1" "1024 4"; do set -- $cfg; FPHIP_BFS_HEAVY=$1 FPHIP_BFS_WG_PER_CU=$2 timeout 300 $P > $O/pruner_hv$1_wg$2.log 2>&1; done
FPHIP_BFS_HEAVY=1024 FPHIP_BFS_WG_PER_CU=1 FPHIP_DEBUG=1 timeout 300 python bench.py --regime pruner --no-cpu --no-gso --no-tour --no-pmc --steps 3 --warmup 1 > $O/pruner_dbg.log 2>&1
( time timeout 600 python -m pytest tests/test_bkzs_gpu.py tests/test_zz_sd_bkz_gpu.py -x -q -m gpu ) > $O/tests_bkzs.log 2>&1
