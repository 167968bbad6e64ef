# (GPU) pruner-regime and main bench over the split parameters of the enumeration host (FPHIP_PHASE_GROWTH, ...)
B="python bench.py --regime pruner --no-cpu --no-gso --no-tour --no-pmc --steps 6 --warmup 1"
M="python bench.py --no-cpu --no-gso --no-tour --no-pmc --steps 2 --warmup 1"
run() { name=$1; shift; env "$@" $B 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name', round(d['ms_per_step'],3), round(d['enum_kernel_ms_per_step'],3), d['parity']['final_norm_not_longer_than_reference'])"; }
runm() { name=$1; shift; env "$@" $M 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$name main', '%.3e'%d['value'], round(d['ms_per_step'],1))"; }
run base X=1
run g6 FPHIP_PHASE_GROWTH=6
run g12 FPHIP_PHASE_GROWTH=12
run g16 FPHIP_PHASE_GROWTH=16
run g24 FPHIP_PHASE_GROWTH=24
run g32 FPHIP_PHASE_GROWTH=32
run g48 FPHIP_PHASE_GROWTH=48
runm base X=1
runm g24 FPHIP_PHASE_GROWTH=24
runm g12 FPHIP_PHASE_GROWTH=12
