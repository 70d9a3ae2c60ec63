#!/bin/bash
# Kernel-trace average of K2 (noc_decode_kernel) over 60 launches on three resident head outputs (development aid; through gpurun, repo root).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/k2_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/k2_trace -o t -- env WHICH=k2 REPS=${REPS:-60} MR_PNP_SO=${MR_PNP_SO:-} python $R/tools/gpu_noc_path.py > /dev/null 2>&1
python - <<'P'
import csv, glob, os
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/k2_trace/**/t_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'noc_decode' in r['Name']: print(f"{r['Name'][:80]:<80} calls {r['Calls']:>4} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f} max {float(r['MaxNs'])/1e3:8.2f}")
P
