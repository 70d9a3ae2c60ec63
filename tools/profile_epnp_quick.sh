#!/bin/bash
# Kernel-trace summary of the EPnP / RANSAC initialiser + LM launches (development aid; run through gpurun from the repo root).
#   bash tools/profile_epnp_quick.sh [filter]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ep_trace
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ep_trace -o t -- env REPS=${REPS:-10} OBJECTS=${OBJECTS:-1024} python $R/tools/gpu_epnp_path.py > /dev/null 2>&1
python - <<'P'
# per launch of the sequence, in issue order, averaged over the calls (a kernel that runs in both rounds appears twice)
import csv, glob, os, collections
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/ep_trace/**/t_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seqs, cur = [], None
for r in rows:
    n = r['Kernel_Name']
    if 'epnp_front_kernel' in n:
        cur = []; seqs.append(cur)
    if cur is not None and ('epnp_' in n or 'pnp_uncert' in n):
        cur.append((n, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp']), int(r['End_Timestamp'])))
seqs = [q for q in seqs if len(q) == len(seqs[-1])][2:]
tot = 0.0
for i in range(len(seqs[0])):
    d = [q[i][1] for q in seqs]
    name = seqs[0][i][0].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{i:2d} {name[:64]:<64} avg {sum(d)/len(d):7.1f} us  min {min(d):7.1f} max {max(d):7.1f}")
    if 'epnp_' in name: tot += sum(d) / len(d)
span = [(q[-1][3] - q[0][2]) / 1e3 for q in seqs]
print('sum of the initialiser kernels %.1f us; first start to last end of a call (initialiser + LM launch) %.1f us (avg of %d calls)' % (tot, sum(span) / len(span), len(seqs)))
P
