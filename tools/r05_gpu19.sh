python -m pytest tests/test_gpu_epnp.py -x -q -m gpu 2>&1 | tail -3
bash tools/epnp_set_valu.sh | tail -4
GROUP=3 DEPTHS=1,4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | cut -c1-60,170-260
GROUP=1 DEPTHS=1 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | cut -c1-60,170-260
