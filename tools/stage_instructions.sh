#!/bin/bash
# Instructions per stage of the fused kernel, two waves per object (development aid): SQ_INSTS_VALU / _SALU / _LDS of the early-exit
# builds (tools/build_variant.sh exit$k -DMR_EXIT_AFTER=$k, k = 1..7: load, mask+list, hypotheses, consensus, refit, LM, cov) and of
# the product build, one 1024-object config-2 launch each; differences = stage costs.  Run through gpurun from the repo root.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/one_launch.py <<'P'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
from monorun_amd import synthetic as syn, PnPLaunch
dev = torch.device('cuda:0')
def dv(a):
    t = torch.from_numpy(np.asarray(a)); d = torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=dev); d.copy_(t); return d
x = [dv(a) for a in syn.pnp_boundary(syn.make_batch(B=1024, seed=1234), planar=True)]
L = PnPLaunch(*x[:6], z_min=0.5, epnp_istd_thres=0.6, epnp_ransac_thres=x[6], inlier_opt_only=True, flags=int(os.environ.get('WAVES', '2')) << 8)
for _ in range(4): L.run()
torch.cuda.synchronize()
P
for k in 1 2 3 4 5 6 7 full; do
    so=$R/monorun_amd/variants/libmr_exit$k.so; [ $k = full ] && so=$R/monorun_amd/libmonorun_pnp.so
    [ -f $so ] || continue
    rm -rf /tmp/si_$k
    MR_PNP_SO=$so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d /tmp/si_$k -o p -- python /tmp/one_launch.py > /tmp/si_$k.log 2>&1
done
python - <<'P'
import csv, glob, collections
names = dict(zip('1234567', ['load', 'mask+list', 'hypotheses', 'consensus', 'refit', 'LM', 'cov']))
prev = collections.defaultdict(float)
print(f'{"stage":<12} ' + ' '.join(f'{c:>22}' for c in ('VALU', 'SALU', 'LDS')) + '   (wave-instructions per object: cumulative / this stage)')
for k in list('1234567') + ['full']:
    f = glob.glob(f'/tmp/si_{k}/**/p_counter_collection.csv', recursive=True)
    if not f: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if 'pnp_uncert_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    row = []
    for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS'):
        v = sum(acc[c]) / max(len(acc[c]), 1) / 1024
        row.append(f'{v:10.0f} /{v - prev[c]:9.0f}'); prev[c] = v
    print(f'{names.get(k, "outputs"):<12} ' + ' '.join(f'{x:>22}' for x in row))
P
