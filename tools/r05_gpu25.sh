python -m pytest tests/test_gpu_epnp.py -x -q -m gpu 2>&1 | tail -3
for g in 4 5 6 8; do NBATCH=24 GROUP=$g DEPTHS=4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | sed 's/(streams found.*): / /'; done
for g in 3 4 5; do echo "== bench --group $g"; for r in 1 2; do python bench.py --steps 20 --warmup 5 --group $g --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.2f M cold %.2f steady %.2f' % (d['value']/1e6, d.get('value_cold',0)/1e6, d['steady_state']['value']/1e6))"; done; done
