#!/bin/bash
# kernel resource usage of one source file (VGPRs, spills, scratch, occupancy): scripts/kres.sh search.hip [name filter] [extra -D flags]
cd /root/repo/kektordb_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DKDB_BUILD $3 \
  -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
pats=(('vgpr',r' VGPRs: (\d+)'),('agpr',r' AGPRs: (\d+)'),('spill',r'VGPRs Spill: (\d+)'),('sspill',r'SGPRs Spill: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('sgpr',r'TotalSGPRs: (\d+)'))
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for k,pat in pats:
        m=re.search(pat,l)
        if m and cur is not None: cur[k]=m.group(1)
dem=subprocess.run(['/usr/bin/c++filt'],input='\n'.join(r['name'] for r in rows),capture_output=True,text=True).stdout.split('\n')
flt=sys.argv[1] if len(sys.argv)>1 else ''
for r,d in zip(rows,dem):
    d=re.sub(r'\(anonymous namespace\)::','',d); d=re.sub(r'\(.*','',d); d=d.replace('void ','')
    if flt in d: print('%-64s vgpr %4s agpr %3s vspill %3s sspill %3s scratch %4s occ %s' % (d[:64], r.get('vgpr'), r.get('agpr'), r.get('spill'), r.get('sspill'), r.get('scratch'), r.get('occ')))
" "$2"
