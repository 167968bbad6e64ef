#!/bin/bash
mkdir -p gpurun_out/exp
B="python bench.py --no-cpu --no-gso --no-tour --no-pmc --steps 4 --warmup 1"
run() { # name, lib, env...
  name=$1; lib=$2; shift 2
  cp exp/$lib fplll_amd/lib/libfplll_hip.so
  env "$@" $B > gpurun_out/exp/$name.log 2> gpurun_out/exp/$name.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/exp/$name.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$name", "%.3e nodes/s"%d["value"], "ms/step %.1f"%d["ms_per_step"], d.get("parity",{}).get("final_norm_equal_to_reference"))
else:
    print("$name FAILED"); print(open("gpurun_out/exp/$name.err").read()[-500:])
PY
}
cp exp/libV3.so fplll_amd/lib/libfplll_hip.so
(timeout 600 python -m pytest tests/test_enum_gpu.py tests/test_enum_multirank_gpu.py tests/test_reference_kats.py -x -q -m gpu 2>&1 | tail -3)
run V3 libV3.so X=1
run V2 libV2.so X=1
run V3_nosplit libV3.so FPHIP_STACK_SPLIT=0
run V3_lvl30 libV3.so FPHIP_MU_GLOBAL_MIN_LEVEL=30
cp exp/libV3.so fplll_amd/lib/libfplll_hip.so
bash exp/pmc.sh
