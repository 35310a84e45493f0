#!/usr/bin/env python3
"""Static instruction mix per kernel from a hipcc -save-temps .s file (gfx950).
usage: tools/asm_mix.py file.s [name-filter]"""
import re, sys, collections
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r'\n(_Z[^\n:]*):[^\n]*\n', s)
for i in range(1, len(parts) - 1, 2):
    name, body = parts[i], parts[i + 1]
    if flt not in name: continue
    body = body.split('.Lfunc_end')[0]
    ins = []
    for l in body.split('\n'):
        t = l.strip()
        if not l.startswith('\t') or not t or t[0] in '.;': continue
        ins.append(t.split()[0])
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    top = ', '.join(f"{k}:{v}" for k, v in c.most_common(14))
    print(f"{name[:70]}\n   total {len(ins)} valu {valu} salu {sum(v for k,v in c.items() if k.startswith('s_'))} cndmask {sum(v for k,v in c.items() if 'cndmask' in k)} s_nop {c.get('s_nop',0)} waitcnt {c.get('s_waitcnt',0)} ds {sum(v for k,v in c.items() if k.startswith('ds_'))} global {sum(v for k,v in c.items() if k.startswith('global_'))}\n   {top}")
