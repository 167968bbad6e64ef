#!/bin/bash
# usage: ana.sh file.hip  → uniformity summary + VALU counts for the <false,false,false> kernel
cd ${ISA_TMP:-/tmp/isa}
F=${1:-enum_kernel.hip}
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-inline-asm --cuda-device-only"
hipcc $FL -S -emit-llvm -o e.ll $F 2>/dev/null || { hipcc $FL -S -emit-llvm -o e.ll $F; exit 1; }
/opt/rocm/lib/llvm/bin/opt -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -passes='print<uniformity>' -disable-output e.ll 2> uni.txt
python3 - <<'PY'
import re
t=open('${ISA_TMP:-/tmp/isa}/uni.txt').read().split("UniformityInfo for function ")
for f in t[1:]:
    name=f.split("'")[1]
    if 'enum_phase_kernelILb0ELb0ELb0' not in name: continue
    cyc=[l for l in f.split('\n') if l.strip().startswith('depth=')]
    print("divergent-exit cycles:",len(cyc))
    print("divergent phis:",len(re.findall(r'DIVERGENT:.*= phi',f)),"divergent br:",len(re.findall(r'DIVERGENT:\s+br ',f)))
PY
hipcc $FL -S -o e.s $F 2>/dev/null
python3 - <<'PY'
import re
lines=open('${ISA_TMP:-/tmp/isa}/e.s').read().split('\n')
st=[i for i,l in enumerate(lines) if l.startswith('_ZN5fphip17enum_phase_kernelILb0ELb0ELb0') and l.rstrip().endswith('dii') is False and ':' in l][0]
en=[i for i in range(st,len(lines)) if 's_endpgm' in lines[i]][0]
body=lines[st:en]
v=sum(1 for l in body if l.strip().startswith('v_')); s=sum(1 for l in body if l.strip().startswith('s_'))
mov=sum(1 for l in body if l.strip().startswith('v_mov'))
rl=sum(1 for l in body if 'v_readlane' in l or 'v_readfirstlane' in l)
print("kernel static: VALU",v,"SALU",s,"v_mov",mov,"readlane",rl)
for l in lines[en:en+400]:
    if '.vgpr_count' in l or '.sgpr_count' in l or 'scratch' in l.lower() and 'size' in l.lower(): pass
PY
grep -A40 "enum_phase_kernelILb0ELb0ELb0" e.s | grep -m3 "NumVgprs\|NumSgprs\|ScratchSize" 
