#!/bin/bash
# round 4, call g: NQ = 3 geometry of the sweep kernel, A/B on one box (libraries built by
# tests/perf/build_sweep_variants.sh): V0 = 4 waves per SIMD in the launch bounds (128 VGPRs), 6 pair slots;
# V1 = 3 waves per SIMD (168 VGPRs: what three blocks per CU allow); V2 = V1 with 13 KB of ring per wave (8 pair slots)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
cp fplll_amd/lib/libfplll_hip.so /tmp/libhead.so
for v in V1 V2; do
  cp exp/lib$v.so fplll_amd/lib/libfplll_hip.so
  ( timeout 200 python -m pytest tests/test_gso_gpu.py tests/test_a_configs_at_size_gpu.py -q -m gpu -k "gso or nq4_size" 2>&1 | tail -1 | sed "s/^/$v parity: /" ) | tee -a $O/roof.log
done
for rep in 1 2 3; do for v in V0 V1 V2; do
  cp exp/lib$v.so fplll_amd/lib/libfplll_hip.so
  timeout 200 python tests/perf/gso_roof.py 8192 2>&1 | tail -1 | sed "s/^/rep $rep $v /"
done; done | tee -a $O/roof.log
cp /tmp/libhead.so fplll_amd/lib/libfplll_hip.so
