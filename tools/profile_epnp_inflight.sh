#!/bin/bash
# Kernel-trace of the reference flow with DEPTH launches in flight: per kernel, the average duration while overlapped (development aid).
#   bash tools/profile_epnp_inflight.sh [depth] [first_round]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=${1:-4}; F=${2:-8}
rm -rf $R/gpurun_out/ep_trace_if
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ep_trace_if -o t -- env DEPTHS=$D GROUP=${GROUP:-5} NBATCH=${NBATCH:-20} python $R/tools/gpu_epnp_inflight.py > $R/gpurun_out/ep_trace_if.log 2>&1
python - <<'P'
import csv, glob, os, collections
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/ep_trace_if/**/t_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'epnp_' in r['Kernel_Name'] or 'pnp_uncert_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 4:]                      # skip warm-up
by = collections.defaultdict(list)
for r in rows:
    by[(r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:48], r['Grid_Size'] if 'Grid_Size' in r else '')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
calls = sum(1 for r in rows if 'epnp_front' in r['Kernel_Name'])
print(f'{calls} calls in {(t1 - t0) / 1e3:.0f} us = {(t1 - t0) / 1e3 / calls:.1f} us per call')
for k, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k[0]:<50} grid {k[1]:>8} n {len(d):5d} avg {sum(d)/len(d):8.1f} us  sum/call {sum(d)/calls:8.1f}')
P
tail -3 $R/gpurun_out/ep_trace_if.log
