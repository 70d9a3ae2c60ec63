python -m pytest tests/test_gpu_epnp.py -x -q -m gpu 2>&1 | tail -15
for g in 1 3; do GROUP=$g DEPTHS=1,4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | sed 's/(streams found.*): / /'; done
