#!/bin/bash
# Kernel-trace summary of the EPnP / RANSAC initialiser + LM launches (development aid; run through gpurun from the repo root).
#   bash tools/profile_epnp_quick.sh [filter]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ep_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ep_trace -o t -- env REPS=10 python $R/tools/gpu_epnp_path.py > /dev/null 2>&1
F=${1:-.} python - <<'P'
import csv, glob, os, re
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/ep_trace/**/t_kernel_stats.csv', recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if not re.search(os.environ['F'], r['Name']): continue
    if 'epnp' in r['Name']: tot += float(r['AverageNs']) / 1e3
    print(f"{r['Name'][:86]:<86} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
print('sum of the epnp kernels: %.1f us' % tot)
P
