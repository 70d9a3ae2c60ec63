#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory (gpurun_out/prof_<tag>) into the tracked profiles/<tag>_* files:
kernel-stats CSVs (verbatim) of both bench regimes, of the NOC path and of the EPnP initialiser, per-kernel PMC means, the
bench lines, host info, and profiles/traffic.json (HBM bytes per launch, corrected as MI355X_MICROARCH.md §HBM prescribes)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

import numpy as np

tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', f'prof_{tag}')
dst = os.path.join(root, 'profiles')
os.makedirs(dst, exist_ok=True)


def find(sub, name):
    hits = glob.glob(os.path.join(src, sub, '**', name), recursive=True)
    return hits[0] if hits else None


def stats_rows(sub):
    f = find(sub, 't_kernel_stats.csv')
    return list(csv.DictReader(open(f))) if f else []


def counters(sub, needle):
    """per-launch means of every counter of the kernels whose name contains `needle` in one --pmc pass"""
    f = find(sub, 'p_counter_collection.csv')
    if not f:
        return {}, None
    acc, meta = collections.defaultdict(list), None
    for r in csv.DictReader(open(f)):
        if needle in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = dict(kernel=r['Kernel_Name'], grid=int(r['Grid_Size']), workgroup=int(r['Workgroup_Size']), vgpr=int(r['VGPR_Count']),
                        sgpr=int(r['SGPR_Count']), scratch=int(r['Scratch_Size']), lds=int(r.get('LDS_Block_Size', 0) or 0))
    return {k: dict(mean=float(np.mean(v)), min=float(np.min(v)), max=float(np.max(v)), launches=len(v)) for k, v in acc.items()}, meta


def kernel_block(prefix, needle, passes=('fetch', 'write', 'sq', 'lds'), trace=None, alg_bytes=None):
    cnt, meta = {}, None
    for p in passes:
        c, m = counters(f'{prefix}_{p}' if not prefix.startswith('pmc') else f'{prefix}_{p}', needle)
        cnt.update(c)
        meta = meta or m
    rows = [r for r in stats_rows(trace) if needle in r['Name']]
    out = dict(kernel=meta, counters=cnt)
    if rows:
        out['rocprof_kernel_avg_us'] = float(rows[0]['AverageNs']) / 1e3
        out['rocprof_calls'] = int(rows[0]['Calls'])
        out['rocprof_min_us'], out['rocprof_max_us'] = float(rows[0]['MinNs']) / 1e3, float(rows[0]['MaxNs']) / 1e3
    if 'FETCH_SIZE' in cnt and 'WRITE_SIZE' in cnt:
        hbm = (2 * cnt['FETCH_SIZE']['mean'] + cnt['WRITE_SIZE']['mean']) * 1024
        out['traffic'] = dict(hbm_bytes_per_launch=hbm, fetch_size_kb_raw=cnt['FETCH_SIZE']['mean'], write_size_kb_raw=cnt['WRITE_SIZE']['mean'],
                              fetch_correction='x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B: MI355X_MICROARCH.md §HBM; calibrated for 4 B/lane and 16 B/lane '
                                               'reads in profiles/r02_fetch_calibration.txt)')
        if alg_bytes:
            out['traffic'].update(algorithmic_bytes_per_launch=alg_bytes, ratio_traffic_over_algorithmic=hbm / alg_bytes)
            if rows:
                out['hbm_roofline'] = dict(achieved_GBps=alg_bytes / (out['rocprof_kernel_avg_us'] * 1e-6) / 1e9, peak_GBps=8000.0,
                                           frac=alg_bytes / (out['rocprof_kernel_avg_us'] * 1e-6) / 1e9 / 8000.0)
    if 'SQ_WAVES' in cnt and cnt['SQ_WAVES']['mean'] > 0:
        w = cnt['SQ_WAVES']['mean']
        d = dict(waves_per_launch=w)
        for k, name in (('SQ_INSTS_VALU', 'valu_insts_per_wave'), ('SQ_INSTS_SALU', 'salu_insts_per_wave'), ('SQ_INSTS_LDS', 'lds_insts_per_wave'), ('SQ_INSTS_VMEM', 'vmem_insts_per_wave')):
            if k in cnt:
                d[name] = cnt[k]['mean'] / w
        if 'SQ_WAVE_CYCLES' in cnt:
            wc = cnt['SQ_WAVE_CYCLES']['mean']
            for k, name in (('SQ_ACTIVE_INST_ANY', 'active_frac_of_wave_cycles'), ('SQ_WAIT_ANY', 'wait_any_frac'), ('SQ_WAIT_INST_ANY', 'wait_inst_frac')):
                if k in cnt:
                    d[name] = cnt[k]['mean'] / wc
            if rows:       # SQ_WAVE_CYCLES counts quad-cycles of resident waves; 1024 SIMDs at the 2.4 GHz peak clock (a lower bound of the residency if the clock was lower)
                d['mean_resident_waves_per_simd'] = 4.0 * wc / (out['rocprof_kernel_avg_us'] * 2400.0 * 1024)
        if 'SQ_LDS_BANK_CONFLICT' in cnt and cnt.get('SQ_LDS_IDX_ACTIVE', {}).get('mean', 0) > 0:
            d['lds_bank_conflict_frac'] = cnt['SQ_LDS_BANK_CONFLICT']['mean'] / cnt['SQ_LDS_IDX_ACTIVE']['mean']
        if 'SQ_INSTS_VALU' in cnt and rows:
            d['valu_min_issue_us_at_4_cycles'] = cnt['SQ_INSTS_VALU']['mean'] * 4.0 / (1024 * 2.4e9) * 1e6
            d['valu_issue_frac_of_kernel_time'] = d['valu_min_issue_us_at_4_cycles'] / out['rocprof_kernel_avg_us']
        out['derived'] = d
    return out


def epnp_block():
    """the initialiser's launches in issue order (a kernel that runs in both rounds appears twice), averaged over the calls of the
    trace, + the SQ / LDS counters of the two eigen launches (told apart by their grid size)"""
    f = find('epnp_trace', 't_kernel_trace.csv')
    if not f:
        return {}
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    seqs, cur = [], None
    for r in rows:
        n = r['Kernel_Name']
        if 'epnp_front_kernel' in n:
            cur = []
            seqs.append(cur)
        if cur is not None and ('epnp_' in n or 'pnp_uncert_kernel' in n):
            cur.append((n, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0))))
    seqs = [q for q in seqs if len(q) == len(seqs[-1])][2:]
    launches = []
    for i in range(len(seqs[0])):
        d = [q[i][1] for q in seqs]
        launches.append(dict(position=i, kernel=seqs[0][i][0], grid_threads=seqs[0][i][2], avg_us=float(np.mean(d)), min_us=float(np.min(d)), max_us=float(np.max(d)), calls=len(d)))
    out = dict(launches=launches, initialiser_sum_us=float(sum(l['avg_us'] for l in launches if 'epnp_' in l['kernel'])),
               lm_launch_us=float(sum(l['avg_us'] for l in launches if 'pnp_uncert_kernel' in l['kernel'])))
    # per-kernel counters (means per launch; a kernel that runs in both rounds: the round-1 launches are the ones with work, the idle
    # round-2 ones pull the mean down — both are listed by name only) and the HBM traffic of ONE call of the flow (all launches)
    per_kernel = {}
    for sub in ('epnp_sq', 'epnp_lds', 'epnp_fetch', 'epnp_write'):
        g = find(sub, 'p_counter_collection.csv')
        if not g:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(g)):
            if 'epnp_' in r['Kernel_Name'] or 'pnp_uncert_kernel' in r['Kernel_Name']:
                acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in acc.items():
            per_kernel.setdefault(k.replace('(anonymous namespace)::', ''), {}).update({c: dict(mean=float(np.mean(v)), sum=float(np.sum(v)), launches=len(v)) for c, v in cs.items()})
    out['counters_per_kernel'] = per_kernel
    calls = None
    for k, cs in per_kernel.items():
        if 'epnp_front_kernel' in k and 'FETCH_SIZE' in cs:
            calls = cs['FETCH_SIZE']['launches']
    if calls:
        fetch = sum(cs['FETCH_SIZE']['sum'] for cs in per_kernel.values() if 'FETCH_SIZE' in cs) / calls
        write = sum(cs['WRITE_SIZE']['sum'] for cs in per_kernel.values() if 'WRITE_SIZE' in cs) / calls
        out['traffic'] = dict(hbm_bytes_per_call=(2 * fetch + write) * 1024, fetch_size_kb_raw_per_call=fetch, write_size_kb_raw_per_call=write, calls=calls,
                              fetch_correction='x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B: MI355X_MICROARCH.md HBM section)',
                              what='all launches of one call of the reference flow (initialiser + LM) on 1024 config-2 objects')
    for k, cs in per_kernel.items():
        if 'SQ_WAVES' in cs and cs['SQ_WAVES']['mean'] > 0:
            w = cs['SQ_WAVES']['mean']
            cs['per_wave'] = {c: cs[c]['mean'] / w for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM') if c in cs}
            if 'SQ_WAVE_CYCLES' in cs and 'SQ_ACTIVE_INST_ANY' in cs:
                cs['active_frac_of_wave_cycles'] = cs['SQ_ACTIVE_INST_ANY']['mean'] / cs['SQ_WAVE_CYCLES']['mean']
    return out


for sub, name in (('trace1', 'kernel_stats'), ('trace4', 'kernel_stats_in_flight'), ('noc_k2_trace', 'k2_kernel_stats'), ('noc_fused_trace', 'fused_kernel_stats'),
                  ('epnp_trace', 'epnp_kernel_stats')):
    f = find(sub, 't_kernel_stats.csv')
    if f:
        shutil.copy(f, os.path.join(dst, f'{tag}_{name}.csv'))
for f in ('lscpu.txt', 'rocminfo.txt'):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f'{tag}_{f}'))
bench = open(os.path.join(src, 'bench.json')).read().strip().split('\n')[-1]
open(os.path.join(dst, f'{tag}_bench.json'), 'w').write(bench + '\n')
if os.path.exists(os.path.join(src, 'bench_240.json')):
    open(os.path.join(dst, f'{tag}_bench_240_steps.json'), 'w').write(open(os.path.join(src, 'bench_240.json')).read().strip().split('\n')[-1] + '\n')
b = json.loads(bench)
alg = b['roofline']['algorithmic_bytes_per_launch']
P, B = 784, 1024
summ = dict(tag=tag,
            single_stream=kernel_block('pmc1', 'pnp_uncert_kernel<float, 4', trace='trace1', alg_bytes=alg),
            in_flight=kernel_block('pmc4', 'pnp_uncert_kernel<float, 2', trace='trace4', alg_bytes=alg),
            k2_noc_decode=kernel_block('noc_k2', 'noc_decode_kernel', passes=('fetch', 'write', 'sq'), trace='noc_k2_trace', alg_bytes=B * P * 48 + B * 80),
            fused_head_to_pose=kernel_block('noc_fused', 'pnp_uncert_kernel', passes=('fetch', 'write', 'sq'), trace='noc_fused_trace',
                                            alg_bytes=B * (P * 20 + 52 + 85 + P + 64 + 24)),
            epnp_stages=epnp_block(),
            bench_kernel_avg_us=b['roofline']['isolated_launch']['kernel_ms_avg'] * 1e3, bench_value=b['value'], bench_single_stream=b.get('single_stream', {}).get('value'))
json.dump(summ, open(os.path.join(dst, f'{tag}_summary.json'), 'w'), indent=1)
t1 = summ['single_stream'].get('traffic')
if t1:
    traffic = dict(tag=tag, **t1, source=f'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 96 --warmup 12 --no-cpu-baseline --no-secondary --in-flight 1` '
                                          f'(tools/profile_round.sh {tag}): the 4-waves-per-object kernel of isolated launches, what roofline.achieved is computed from',
                   in_flight_kernel=summ['in_flight'].get('traffic'))
    json.dump(traffic, open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
for k in ('single_stream', 'in_flight', 'k2_noc_decode', 'fused_head_to_pose'):
    blk = summ[k]
    print(k, 'avg us', blk.get('rocprof_kernel_avg_us'), 'calls', blk.get('rocprof_calls'), 'traffic ratio', (blk.get('traffic') or {}).get('ratio_traffic_over_algorithmic'),
          'roofline', (blk.get('hbm_roofline') or {}).get('frac'), 'derived', json.dumps(blk.get('derived')))
ep = summ['epnp_stages']
for l in ep.get('launches', []):
    print('epnp launch %2d %-70s avg us %7.1f' % (l['position'], l['kernel'].replace('(anonymous namespace)::', '')[:70], l['avg_us']))
print('initialiser sum us %.1f, LM launch %.1f' % (ep.get('initialiser_sum_us', 0), ep.get('lm_launch_us', 0)), 'traffic per call', json.dumps(ep.get('traffic')))
if ep.get('traffic'):
    json.dump(dict(tag=tag, **ep['traffic'], algorithmic_bytes_per_call=alg, ratio_traffic_over_algorithmic=ep['traffic']['hbm_bytes_per_call'] / alg,
                   source=f'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gpu_epnp_path.py (tools/profile_round.sh {tag})'),
              open(os.path.join(dst, f'{tag}_epnp_traffic.json'), 'w'), indent=1)
with open(os.path.join(dst, f'{tag}_epnp_launches.txt'), 'w') as f:
    f.write('the launches of one call of the reference flow on 1024 config-2 objects (batch 0), rocprofv3 kernel trace, averaged over the calls of the trace\n')
    for l in ep.get('launches', []):
        f.write('%2d %-72s avg %7.1f us  min %7.1f  max %7.1f\n' % (l['position'], l['kernel'].replace('(anonymous namespace)::', '').replace('void ', '')[:72], l['avg_us'], l['min_us'], l['max_us']))
    f.write('initialiser sum %.1f us, LM launch %.1f us\n' % (ep.get('initialiser_sum_us', 0), ep.get('lm_launch_us', 0)))
print('bench events avg us', summ['bench_kernel_avg_us'], 'value', b['value'], 'single_stream', summ['bench_single_stream'])
