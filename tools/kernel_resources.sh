#!/bin/bash
# Register / spill / scratch / occupancy table of every kernel in the library (development aid; profiles/rNN_kernel_resources.txt)
#   tools/kernel_resources.sh [name-filter-regex] [extra hipcc flags...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
FILTER=${1:-.}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I $ROOT/include --cuda-device-only -c "$@" \
    -Rpass-analysis=kernel-resource-usage $ROOT/monorun_amd/csrc/monorun_pnp.hip -o /dev/null 2>&1 | \
python3 -c "
import re, sys
cur = {}
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: cur = {'name': m.group(1)}
    for k, pat in (('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('sgpr', r' SGPRs: (\d+)'), ('ss', r'SGPRs Spill: (\d+)'), ('vs', r'VGPRs Spill: (\d+)'), ('scr', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, l)
        if m: cur[k] = int(m.group(1))
    if 'lds' in cur and 'name' in cur:
        if re.search(r'$FILTER', cur['name']):
            print(f\"{cur['name']:<75} VGPRs {cur.get('vgpr',0):3d} AGPRs {cur.get('agpr',0):3d} SGPRs {cur.get('sgpr',0):3d} sgpr-spill {cur.get('ss',0):3d} vgpr-spill {cur.get('vs',0):3d} scratch {cur.get('scr',0):4d} waves/SIMD {cur.get('occ',0)}\")
        cur = {}
"
