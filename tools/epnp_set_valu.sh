#!/bin/bash
# wave-instruction counts of every launch of ONE LAUNCH SET of GROUP (default 5) calls (development aid; profiles/rNN_epnp_valu_per_launch_set.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ep_set_valu; rm -rf $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o p -- env REPS=6 GROUP=${GROUP:-5} python $R/tools/gpu_epnp_set_path.py > $O.log 2>&1
python - <<'P'
import csv, glob, os, collections
G = int(os.environ.get('GROUP', 5))
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/ep_set_valu/**/p_counter_collection.csv', recursive=True)[0]
by = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    by.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name']})[r['Counter_Name']] = float(r['Counter_Value'])
d = [v for k, v in sorted(by.items()) if 'epnp_' in v['name'] or 'pnp_uncert' in v['name']]
calls, cur = [], None
for v in d:
    if 'epnp_front' in v['name']: cur = []; calls.append(cur)
    if cur is not None: cur.append(v)
calls = [c for c in calls if len(c) == len(calls[-1])][1:]
tot = 0
for i in range(len(calls[0])):
    n = calls[0][i]['name'].replace('(anonymous namespace)::', '').replace('void ', '')[:50]
    g = lambda key: sum(c[i].get(key, 0) for c in calls) / len(calls)
    print(f"{i} {n:<52} waves {g('SQ_WAVES'):7.0f}  VALU {g('SQ_INSTS_VALU')/1e6:6.2f} M  SALU {g('SQ_INSTS_SALU')/1e6:5.2f} M  per call {g('SQ_INSTS_VALU')/G/1e6:6.2f} M")
    tot += g('SQ_INSTS_VALU')
print(f'total VALU per launch set of {G} calls {tot/1e6:.1f} M = {tot/G/1e6:.1f} M per call')
P
