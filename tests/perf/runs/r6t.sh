#!/bin/bash
# tuning sweep of the breadth-first stage on the pruner-regime calls (bench --regime pruner: 24 calls per step)
mkdir -p gpurun_out/r6t
B="python bench.py --regime pruner --no-cpu --no-gso --no-tour --no-pmc --no-batch --steps 3 --warmup 1"
run() {
  echo "== $*"
  env "$@" timeout 120 $B 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('  value %.3e  ms/step %.3f  parity %s' % (d['value'], d['ms_per_step'], d.get('parity')))"
}
{
run FPHIP_NOP=1
run FPHIP_BFS_SINGLE_MAX=64
run FPHIP_BFS_SINGLE_MAX=1024
run FPHIP_BFS_HEAVY=4096
run FPHIP_BFS_HEAVY=16384
run FPHIP_BFS_TASKS=16384
run FPHIP_BFS_WG_PER_CU=2
run FPHIP_BFS_FLOOR=12
run FPHIP_NOP=1
} > gpurun_out/r6t/sweep.log 2>&1
cat gpurun_out/r6t/sweep.log
