set -x
O=gpurun_out/r3o; mkdir -p $O
B="python bench.py --steps 5 --warmup 2 --no-cpu --no-gso --no-tour --no-pmc"
run() { name=$1; shift; ( env "$@" timeout 200 $B ) > $O/$name.log 2> $O/$name.err; python - $O/$name.log $name <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
print(sys.argv[2], "%.4e"%json.loads(l[-1])['value'] if l else 'FAILED', json.loads(l[-1])['parity']['final_norm_equal_to_reference'] if l else '')
PY
}
run default X=1
run split32 FPHIP_STACK_SPLIT=32
run split36 FPHIP_STACK_SPLIT=36
run split30 FPHIP_STACK_SPLIT=30
run wpb4 FPHIP_WAVES_PER_BLOCK=4
run wpb1 FPHIP_WAVES_PER_BLOCK=1
run mulds FPHIP_MU_GLOBAL_MIN_LEVEL=99
