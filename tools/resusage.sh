#!/bin/bash
# developer aid: per-kernel VGPR/SGPR/LDS/scratch/occupancy of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage)
# usage: tools_resusage.sh file.hip  -> one line per kernel: name VGPRs SGPRs scratch LDS occupancy
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -ffp-contract=off -std=c++17 -I'$(dirname "$0")/../include' -c "$1" -o /tmp/scratch/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur={}
for line in sys.stdin:
    m=re.search(r'remark: +(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs|wavefront size): (\S+)',line) or re.search(r':\d+:\d+: +(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)',line)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k in('Function Name','Name'):
        if cur: print(cur)
        cur={'name':v}
    else: cur[k.split(' ')[0]]=v
if cur: print(cur)
"
