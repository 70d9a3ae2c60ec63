#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory (gpurun_out/prof_<tag>) into the tracked
profiles/<tag>_* files: kernel stats CSV (verbatim), per-kernel PMC means, the bench line, host info,
and profiles/traffic.json (HBM bytes per launch, corrected as MI355X_MICROARCH.md §HBM prescribes)."""
import collections
import csv
import json
import os
import shutil
import sys

import numpy as np

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', f'prof_{tag}')
dst = os.path.join(root, 'profiles')
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, 'trace', 't_kernel_stats.csv'), os.path.join(dst, f'{tag}_kernel_stats.csv'))
for f in ('lscpu.txt', 'rocminfo.txt'):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f'{tag}_{f}'))
bench = open(os.path.join(src, 'bench.json')).read().strip().split('\n')[-1]
open(os.path.join(dst, f'{tag}_bench.json'), 'w').write(bench + '\n')
counters, meta = {}, None
for d in sorted(os.listdir(src)):
    f = os.path.join(src, d, 'p_counter_collection.csv')
    if not os.path.exists(f):
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'pnp_uncert_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = dict(kernel=r['Kernel_Name'], grid=int(r['Grid_Size']), workgroup=int(r['Workgroup_Size']),
                        vgpr=int(r['VGPR_Count']), sgpr=int(r['SGPR_Count']), scratch=int(r['Scratch_Size']))
    for k, v in acc.items():
        counters[k] = dict(mean=float(np.mean(v)), min=float(np.min(v)), max=float(np.max(v)), launches=len(v), pass_dir=d)
b = json.loads(bench)
fetch_kb = counters['FETCH_SIZE']['mean']; write_kb = counters['WRITE_SIZE']['mean']
# gfx950: FETCH_SIZE tallies 128-B requests at 64 B for coalesced streaming reads -> double it.  The guide states this for
# 16 B/lane reads; the kernel's HBM reads are one dword per lane (load_records), so the factor was calibrated for that width too:
# profiles/r02_fetch_calibration.txt (tools/ubench/fetch_calib.hip: counter / true bytes = 0.5000 at 4 B/lane and at 16 B/lane).
# WRITE_SIZE is used as reported.
hbm = (2 * fetch_kb + write_kb) * 1024
traffic = dict(tag=tag, hbm_bytes_per_launch=hbm, fetch_size_kb_raw=fetch_kb, write_size_kb_raw=write_kb,
               fetch_correction='x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B; MI355X_MICROARCH.md §HBM; calibrated for 4 B/lane reads in profiles/r02_fetch_calibration.txt)',
               source=f'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary` (tools/profile_round.sh {tag})',
               algorithmic_bytes_per_launch=b['roofline']['algorithmic_bytes_per_launch'],
               ratio_traffic_over_algorithmic=hbm / b['roofline']['algorithmic_bytes_per_launch'])
json.dump(traffic, open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
stats = [r for r in csv.DictReader(open(os.path.join(src, 'trace', 't_kernel_stats.csv'))) if 'pnp_uncert_kernel' in r['Name']][0]
summ = dict(tag=tag, kernel=meta, rocprof_kernel_avg_us=float(stats['AverageNs']) / 1e3, rocprof_calls=int(stats['Calls']),
            bench_kernel_avg_us=b['roofline']['kernel_ms_avg'] * 1e3, counters=counters, traffic=traffic,
            derived=dict(valu_insts_per_wave=counters['SQ_INSTS_VALU']['mean'] / counters['SQ_WAVES']['mean'],
                         salu_insts_per_wave=counters['SQ_INSTS_SALU']['mean'] / counters['SQ_WAVES']['mean'],
                         lds_insts_per_wave=counters['SQ_INSTS_LDS']['mean'] / counters['SQ_WAVES']['mean'],
                         vmem_insts_per_wave=counters['SQ_INSTS_VMEM']['mean'] / counters['SQ_WAVES']['mean'],
                         active_frac_of_wave_cycles=counters['SQ_ACTIVE_INST_ANY']['mean'] / counters['SQ_WAVE_CYCLES']['mean'],
                         wait_any_frac=counters['SQ_WAIT_ANY']['mean'] / counters['SQ_WAVE_CYCLES']['mean'],
                         wait_inst_frac=counters['SQ_WAIT_INST_ANY']['mean'] / counters['SQ_WAVE_CYCLES']['mean'],
                         lds_bank_conflict_frac=counters['SQ_LDS_BANK_CONFLICT']['mean'] / counters['SQ_LDS_IDX_ACTIVE']['mean'],
                         waves_per_launch=counters['SQ_WAVES']['mean'],
                         # SQ_WAVE_CYCLES counts quad-cycles of resident waves (MI355X_MICROARCH.md); divided by the launch's
                         # cycles on 1024 SIMDs (at the 2.4 GHz peak clock: a LOWER bound of the occupancy if the clock was lower)
                         mean_resident_waves_per_simd=4.0 * counters['SQ_WAVE_CYCLES']['mean'] / (float(stats['AverageNs']) / 1e3 * 2400.0 * 1024)))
json.dump(summ, open(os.path.join(dst, f'{tag}_summary.json'), 'w'), indent=1)
print(json.dumps(summ['derived'], indent=1)); print(json.dumps(traffic, indent=1))
print('rocprof avg us', summ['rocprof_kernel_avg_us'], 'bench events avg us', summ['bench_kernel_avg_us'])
