#!/bin/bash
# kernel resource usage of one .hip file: tools/kres.sh lvi-exc_amd/csrc/lvx_eval.hip [filter]
cd "$(dirname "$1")" && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -I../../include -c "$(basename "$1")" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | python3 -c "
import sys,re
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    m=re.search(r'remark:\s+([A-Za-z /\[\]]+): (\d+)',l)
    if m and cur is not None: cur[m.group(1).strip()]=m.group(2)
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()[:70]
    print(f\"{n:70s} V {r.get('VGPRs','?'):>3} A {r.get('AGPRs','?'):>3} scr {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?')} vspill {r.get('VGPRs Spill','?'):>3} sspill {r.get('SGPRs Spill','?'):>3} lds {r.get('LDS Size [bytes/block]','?')}\")
"
