#!/bin/bash
# Build a named variant of libmonorun_pnp.so for A/B measurements (development aid):
#   tools/build_variant.sh <tag> [extra hipcc flags...]   ->  monorun_amd/variants/libmr_<tag>.so  (+ resource usage of the fp32 PnP kernels)
# Select it at run time with MR_PNP_SO=monorun_amd/variants/libmr_<tag>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; shift
mkdir -p $ROOT/monorun_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I $ROOT/include "$@" -Rpass-analysis=kernel-resource-usage \
    $ROOT/monorun_amd/csrc/monorun_pnp.hip -o $ROOT/monorun_amd/variants/libmr_$TAG.so 2> /tmp/res_$TAG.txt || { tail -30 /tmp/res_$TAG.txt; exit 1; }
python3 - "$TAG" <<'PY'
import re, sys
t = open(f'/tmp/res_{sys.argv[1]}.txt').read()
for blk in re.split(r'remark: [^\n]*Function Name: ', t)[1:]:
    name = blk.split()[0]
    if 'pnp_uncert_kernelIf' not in name and 'pnp_noc' not in name and 'pnp6' not in name:
        continue
    g = lambda k: (re.search(k + r': (\d+)', blk) or [None, '?'])[1]
    print(f'{name[:60]:60s} VGPRs {g("VGPRs")} AGPRs {g("AGPRs")} SGPRs {g("SGPRs")} sgpr-spill {g("SGPRs Spill")} vgpr-spill {g("VGPRs Spill")} scratch {g("ScratchSize .bytes/lane.")} occ {g("Occupancy .waves/SIMD.")} LDS {g("LDS Size .bytes/block.")}')
PY
