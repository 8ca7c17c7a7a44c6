#!/bin/bash
# Dev tool: VGPR / scratch / occupancy per kernel of restir_rt.hip
cd "$(dirname "$0")/../cedec_2024_rt_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero -Wno-unused-value -c restir_rt.hip -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; d={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d[cur]={}; continue
    m=re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)',l)
    if m and cur: d[cur][m.group(1).strip()]=m.group(2)
for k,v in d.items():
    if 'rocprim' in k or 'k_' not in k: continue
    print('%-60s VGPR %3s SGPR %3s scratch %4s occ %s LDS %s' % (k[:60], v.get('VGPRs'), v.get('TotalSGPRs'), v.get('ScratchSize'), v.get('Occupancy'), v.get('LDS Size')))
"
