#!/bin/bash
# resource usage of every kernel of one generated unit: name <template arguments> VGPR AGPR scratch occupancy LDS.   tools/unit_resources.sh u_20_8_30
# resource usage of every kernel in a generated unit: name-template-args VGPR AGPR scratch occupancy
mkdir -p /tmp/ru
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Rpass-analysis=kernel-resource-usage -c /root/repo/tinympc_amd/csrc/_gen/$1.hip -o /tmp/ru/$1.o 2>&1 | python3 -c "
import sys,re
cur=None
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m:
        cur=m.group(1); vals={}
    for k in ('VGPRs','AGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(k+r': (\d+)',ln)
        if m: vals[k.split()[0]]=m.group(1)
    if 'LDS Size' in ln and cur:
        import subprocess
        name=subprocess.run(['c++filt',cur],capture_output=True,text=True).stdout.strip()
        name=re.sub(r'tinympc_amd::|\(tinympc_amd::SolveArgs\)|void ','',name)
        print(name, vals)
"
