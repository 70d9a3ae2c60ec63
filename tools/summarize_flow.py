#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag> (tools/profile_flow.sh) into the tracked profiles/<tag>_* files (also run on the GPU box between the PMC passes and
the bench lines, so that the lines replay the round's own counters)."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, 'gpurun_out', f'prof_{tag}'), os.path.join(root, 'profiles')
find = lambda sub, name: (glob.glob(os.path.join(src, sub, '**', name), recursive=True) or [None])[0]
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '')
for name, to in (('bench.json', 'bench.json'), ('bench_240.json', 'bench_240_steps.json'), ('bench_stress.json', 'bench_stress.json'), ('lscpu.txt', 'lscpu.txt'), ('rocminfo.txt', 'rocminfo.txt')):
    if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 0:
        shutil.copy(os.path.join(src, name), os.path.join(dst, f'{tag}_{to}'))
for sub, to in (('trace_bench', 'kernel_stats.csv'), ('trace_k0', 'kernel_stats_k0_fast_mode.csv'), ('trace_call', 'kernel_stats_one_call_at_a_time.csv')):
    f = find(sub, 't_kernel_stats.csv')
    if f:
        shutil.copy(f, os.path.join(dst, f'{tag}_{to}'))
# one call at a time: the launches in issue order
def launches_of(sub, objects, to):
    f = find(sub, 't_kernel_trace.csv')
    lines = []
    if not f:
        return
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    seqs, cur = [], None
    for r in rows:
        n = r['Kernel_Name']
        if 'epnp_front' in n:
            cur = []; seqs.append(cur)
        if cur is not None and ('epnp_' in n or 'pnp_uncert_' in n):
            cur.append((short(n), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    seqs = [q for q in seqs if len(q) == len(seqs[-1])][2:]
    lines.append(f'the launches of one call of the reference flow on {objects} config-2 objects (batch 0), rocprofv3 kernel trace, averaged over the calls of the trace')
    for i in range(len(seqs[0])):
        d = [q[i][1] for q in seqs]
        lines.append(f'{i:2d} {seqs[0][i][0][:70]:<70} avg {sum(d)/len(d):7.1f} us  min {min(d):7.1f}  max {max(d):7.1f}')
    span = [(q[-1][3] - q[0][2]) / 1e3 for q in seqs]
    lines.append(f'first start to last end of a call {sum(span)/len(span):.1f} us (avg of {len(seqs)} calls)')
    open(os.path.join(dst, f'{tag}_{to}'), 'w').write('\n'.join(lines) + '\n')
launches_of('trace_call', 1024, 'epnp_launches.txt')
launches_of('trace_call_B100', 100, 'epnp_launches_B100.txt')
def per_kernel(sub):
    f = find(sub, 'p_counter_collection.csv')
    by = collections.OrderedDict()
    if f:
        for r in csv.DictReader(open(f)):
            by.setdefault(int(r['Dispatch_Id']), {'name': short(r['Kernel_Name'])})[r['Counter_Name']] = float(r['Counter_Value'])
    d = [v for k, v in sorted(by.items()) if 'epnp_' in v['name'] or 'pnp_uncert_' in v['name']]
    calls, cur = [], None
    for v in d:
        if 'epnp_front' in v['name']:
            cur = []; calls.append(cur)
        if cur is not None:
            cur.append(v)
    return [c for c in calls if calls and len(c) == len(calls[-1])][1:]
sq, lds = per_kernel('pmc_sq'), per_kernel('pmc_lds')
if sq:
    out, tot = ['wave-instruction counts of every launch of one call of the reference flow (batch 0), rocprofv3 --pmc, averaged over the calls'], 0
    for i in range(len(sq[0])):
        g = lambda key, cc=sq: sum(c[i].get(key, 0) for c in cc) / len(cc)
        out.append(f"{i} {sq[0][i]['name'][:46]:<46} waves {g('SQ_WAVES'):6.0f}  VALU {g('SQ_INSTS_VALU')/1e6:6.2f} M  SALU {g('SQ_INSTS_SALU')/1e6:5.2f} M  LDS {g('SQ_INSTS_LDS')/1e6:5.2f} M  "
                   f"VALU/wave {g('SQ_INSTS_VALU')/max(g('SQ_WAVES'),1):7.0f}  issuing {g('SQ_ACTIVE_INST_VALU')/max(g('SQ_WAVE_CYCLES'),1):.2f} of wave cycles" +
                   (f"  LDS bank-conflict fraction {sum(c[i].get('SQ_LDS_BANK_CONFLICT',0) for c in lds)/max(sum(c[i].get('SQ_LDS_IDX_ACTIVE',0) for c in lds),1):.2f}" if lds and len(lds[0]) == len(sq[0]) else ''))
        tot += g('SQ_INSTS_VALU')
    out.append(f'total VALU per call {tot/1e6:.1f} M  (at 4 cycles each on 1024 SIMDs at 2.4 GHz: {tot*4/1024/2.4e3:.1f} us of issue time)')
    open(os.path.join(dst, f'{tag}_epnp_valu_per_launch.txt'), 'w').write('\n'.join(out) + '\n')
    gi = lambda i, key: sum(c[i].get(key, 0) for c in sq) / len(sq)
    json.dump({'tag': tag, 'valu_insts_per_call': tot, 'what': 'SQ_INSTS_VALU summed over the launches of one call of the reference flow (1024 config-2 objects, batch 0, one call at a time)',
               'per_launch': [{'kernel': sq[0][i]['name'][:60], 'waves': gi(i, 'SQ_WAVES'), 'valu': gi(i, 'SQ_INSTS_VALU'), 'salu': gi(i, 'SQ_INSTS_SALU'), 'lds': gi(i, 'SQ_INSTS_LDS'),
                               'valu_issue_fraction_of_wave_cycles': gi(i, 'SQ_ACTIVE_INST_VALU') / max(gi(i, 'SQ_WAVE_CYCLES'), 1)} for i in range(len(sq[0]))],
               'source': f'rocprofv3 --pmc SQ_* pass of tools/gpu_epnp_path.py (tools/profile_flow.sh {tag})'}, open(os.path.join(dst, f'{tag}_epnp_valu_per_launch.json'), 'w'), indent=1)
def traffic_of(pre, alg, what, to):
    fe, wr = per_kernel(pre + 'FETCH_SIZE'), per_kernel(pre + 'WRITE_SIZE')
    if not (fe and wr):
        return
    fk = sum(sum(v.get('FETCH_SIZE', 0) for v in c) for c in fe) / len(fe)
    wk = sum(sum(v.get('WRITE_SIZE', 0) for v in c) for c in wr) / len(wr)
    per = [{'kernel': fe[0][i]['name'][:60], 'fetch_kb_raw': sum(c[i].get('FETCH_SIZE', 0) for c in fe) / len(fe), 'write_kb_raw': sum(c[i].get('WRITE_SIZE', 0) for c in wr) / len(wr)} for i in range(len(fe[0]))]
    json.dump({'tag': tag, 'hbm_bytes_per_call': (2 * fk + wk) * 1024, 'fetch_size_kb_raw_per_call': fk, 'write_size_kb_raw_per_call': wk, 'calls': len(fe),
               'fetch_correction': 'x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B: MI355X_MICROARCH.md HBM section)',
               'what': what, 'algorithmic_bytes_per_call': alg,
               'ratio_traffic_over_algorithmic': (2 * fk + wk) * 1024 / alg, 'per_launch': per,
               'source': f'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gpu_epnp_path.py (tools/profile_flow.sh {tag})'}, open(os.path.join(dst, f'{tag}_{to}'), 'w'), indent=1)
traffic_of('pmc_', 1024 * 22877, 'all launches of one call of the reference flow (initialiser + LM) on 1024 config-2 objects', 'epnp_traffic.json')
traffic_of('pmc_stress_', 8192 * 47181, 'all launches of one call of the reference flow on the config-5 shard (8192 x 56x56, fp16 storage)', 'epnp_traffic_stress.json')
print('summarised into profiles/' + tag + '_*')
