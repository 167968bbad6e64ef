import re,sys
kern=sys.argv[1] if len(sys.argv)>1 else '_ZN5fphip17enum_phase_kernelILb0ELb0ELb0'
lines=open('/tmp/isa/e.s').read().split('\n')
st=[i for i,l in enumerate(lines) if l.startswith(kern) and ':' in l][0]
en=[i for i in range(st,len(lines)) if 's_endpgm' in lines[i]][0]
body=lines[st:en]
open('/tmp/isa/k1.s','w').write('\n'.join(body))
cur=None;stats=[]
for i,l in enumerate(body):
    m=re.match(r'^(\.LBB[0-9_]+):',l)
    if m or l.startswith('; %bb.'):
        mm=re.search(r'Depth=(\d)',l)
        cur=[(m.group(1) if m else l.split()[1]),i,0,0,0,mm.group(1) if mm else '-', 'Hdr' if 'Header:' in l else ''];stats.append(cur)
        continue
    s=l.strip()
    if cur is None: continue
    if s.startswith('v_'):
        cur[2]+=1
        if s.startswith('v_mov'): cur[4]+=1
    elif s.startswith('s_'): cur[3]+=1
for n,i,v,s,mv,d,h in stats:
    if d=='3': print(f"{n:12s} line {i:5d} VALU {v:3d} (mov {mv:2d}) SALU {s:3d} depth {d} {h}")
