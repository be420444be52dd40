#!/bin/bash
# Builder tool: registers / scratch / LDS / occupancy of every kernel in a .hip source (compiler's own report), e.g.
#   bash tools/kernel_resources.sh conv_clx.hip [name filter]
cd "$(dirname "$0")/../sbv2-api_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres_$$.o 2>&1 |
  python3 -c "
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ''
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m: cur = m.group(1); rows[cur] = {}
    for key in ('VGPRs', 'AGPRs', 'ScratchSize \[bytes/lane\]', 'Occupancy \[waves/SIMD\]', 'LDS Size \[bytes/block\]', 'SGPRs'):
        m = re.search(key + r': (\d+)', line)
        if m and cur: rows[cur][key.split(' ')[0]] = int(m.group(1))
import subprocess
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    if flt in name: print(f\"{name[:110]:110s} vgpr {v.get('VGPRs')} agpr {v.get('AGPRs')} sgpr {v.get('SGPRs')} scratch {v.get('ScratchSize')} occ {v.get('Occupancy')}\")
" "$2"
rm -f /tmp/kres_$$.o
