python -m pytest tests/test_gpu_parity.py tests/test_gpu_epnp.py -x -q -m gpu -k "not bench_line" 2>&1 | tail -3
for i in 1 2; do python bench.py --flow k0 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k0', d['value']/1e6, d['steady_state']['value']/1e6, d['single_stream']['value']/1e6)"; done
GROUP=3 DEPTHS=1,4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids
