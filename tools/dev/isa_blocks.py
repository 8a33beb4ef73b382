#!/usr/bin/env python3
"""Per basic block instruction mix of the ls_mq kernels in a --save-temps .s file (developer scratch)."""
import re, sys
s=open(sys.argv[1]).read()
want=[tuple(x.split(',')) for x in sys.argv[2:]]
parts=re.split(r'\n(?=_Z12ls_mq_kernel\w+: )', s)
for f in parts[1:]:
    name=f.split(':')[0]
    t=re.search(r'ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E',name).groups()
    if want and t not in want: continue
    end=f.find('.Lfunc_end')
    body=f[:end if end>0 else len(f)]
    lines=body.split('\n')
    blocks=[];cur=[];lab='entry'
    for ln in lines:
        m=re.match(r'^(\.LBB\S+):',ln)
        if m:
            blocks.append((lab,cur));cur=[];lab=m.group(1)
        else: cur.append(ln)
    blocks.append((lab,cur))
    print(t)
    for lab,b in blocks:
        mf=sum('v_mfma' in x for x in b)
        sc=sum('scratch_' in x for x in b)
        if mf==0 and sc==0: continue
        acr=sum('v_accvgpr_read' in x for x in b); acw=sum('v_accvgpr_write' in x for x in b)
        ds=sum(re.search(r'\bds_',x) is not None for x in b)
        gl=sum('global_load' in x for x in b); va=sum(re.match(r'\s+v_',x) is not None for x in b)
        wc=sum('s_waitcnt' in x for x in b); mov=sum('v_mov_b32' in x for x in b)
        nins=sum(re.match(r'\s+[a-z]',x) is not None for x in b)
        print(f'  {lab}: insts={nins} mfma={mf} accread={acr} accwrite={acw} scratch={sc} ds={ds} gload={gl} valu={va} waitcnt={wc} vmov={mov}')
