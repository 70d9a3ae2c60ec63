#!/bin/bash
# VALU / SALU / LDS instruction counts of every launch of one call of the reference flow, in issue order (development aid).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ep_valu; rm -rf $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o p -- env REPS=6 python $R/tools/gpu_epnp_path.py > $O.log 2>&1
python - <<'P'
import csv, glob, os, collections
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/ep_valu/**/p_counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.OrderedDict()
for r in rows:
    k = int(r['Dispatch_Id'])
    by.setdefault(k, {'name': r['Kernel_Name']})[r['Counter_Name']] = float(r['Counter_Value'])
d = [v for k, v in sorted(by.items()) if 'epnp_' in v['name'] or 'pnp_uncert_' in v['name']]
# group into calls by front kernel
calls, cur = [], None
for v in d:
    if 'epnp_front' in v['name']: cur = []; calls.append(cur)
    if cur is not None: cur.append(v)
calls = [c for c in calls if len(c) == len(calls[-1])][1:]
tot = 0
for i in range(len(calls[0])):
    n = calls[0][i]['name'].replace('(anonymous namespace)::', '').replace('void ', '')[:44]
    g = lambda key: sum(c[i].get(key, 0) for c in calls) / len(calls)
    print(f"{i} {n:<46} waves {g('SQ_WAVES'):7.0f}  VALU {g('SQ_INSTS_VALU')/1e6:6.2f} M  SALU {g('SQ_INSTS_SALU')/1e6:5.2f} M  LDS {g('SQ_INSTS_LDS')/1e6:5.2f} M  VALU/wave {g('SQ_INSTS_VALU')/max(g('SQ_WAVES'),1):7.0f}")
    tot += g('SQ_INSTS_VALU')
print(f'total VALU per call {tot/1e6:.1f} M')
P
