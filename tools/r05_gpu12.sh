mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wave_counts or bench_line" 2>&1 | tail -4
python bench.py --workload stress > gpurun_out/r05/bench_stress.json 2> gpurun_out/r05/bench_stress.err; echo rc=$?; tail -3 gpurun_out/r05/bench_stress.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05/bench_stress.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('dominant_kernel'), d.get('cpu_baseline'), d['config']['flow'][:30])
P
python bench.py --workload stress --flow k0 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k0 stress', d['value'], d['roofline']['frac'])"
