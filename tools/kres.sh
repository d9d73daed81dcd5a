#!/bin/bash
# usage: tools/kres.sh file.hip  -> one line per kernel: name VGPRs spill occupancy LDS
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    if 'error' in l or 'warning' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        name=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip()
        cur={'name':re.sub(r'\(.*','',name)[-70:]}; rows.append(cur); continue
    for k in ('VGPRs','AGPRs','VGPRs Spill','SGPRs','Occupancy [waves/SIMD]','LDS Size [bytes/block]','ScratchSize [bytes/lane]'):
        m=re.search(r'remark:\s+'+re.escape(k)+r': (\d+)',l)
        if m and cur is not None: cur[k]=m.group(1)
for r in rows:
    print('%-72s v=%s a=%s spill=%s sgpr=%s occ=%s lds=%s scratch=%s'%(r['name'],r.get('VGPRs'),r.get('AGPRs'),r.get('VGPRs Spill'),r.get('SGPRs'),r.get('Occupancy [waves/SIMD]'),r.get('LDS Size [bytes/block]'),r.get('ScratchSize [bytes/lane]')))
"
