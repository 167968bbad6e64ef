import re,sys
t=open('/tmp/isa/uni.txt').read().split("UniformityInfo for function ")
f=[x for x in t[1:] if 'enum_phase_kernelILb0ELb0ELb0' in x.split("'")[1]][0]
lines=f.split('\n')
div=set()
defs={}
blk=None
for l in lines:
    if l.startswith('BLOCK'): blk=l.split()[1]
    m=re.match(r'\s*(DIVERGENT:)?\s*(%\d+) = (.*)',l)
    if m:
        defs[m.group(2)]=(blk,bool(m.group(1)),m.group(3))
        if m.group(1): div.add(m.group(2))
# a divergent def none of whose operands are divergent = seed
for v,(b,d,rhs) in defs.items():
    if not d: continue
    ops=set(re.findall(r'%\d+',rhs))
    if not (ops & div):
        print('SEED blk',b,v,'=',rhs[:150])
