mkdir -p gpurun_out/r05
python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench.json 2> gpurun_out/r05/bench.err; echo rc=$?
tail -5 gpurun_out/r05/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'frac', d['roofline']['frac'])
print('steady', d['steady_state']['value'], 'single', d['single_stream']['value'])
print('split', d['reference_flow']['launch_split'])
print('cold', d['config']['prewarm'].get('window_before',{}).get('value'))
print('dominant', d['roofline'].get('dominant_kernel'))
k=d.get('k0_fast_mode',{})
print('k0', k.get('value'), k.get('error'), (k.get('steady_state') or {}).get('value'))
print('cpu', d.get('cpu_baseline'), d.get('speedup_vs_cpu_1thread'))
print('cpu all', d.get('cpu_baseline_all_cores'))
P
python bench.py --flow k0 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k0 alone', d['value'], d['steady_state']['value'])"
