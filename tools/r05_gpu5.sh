python -m pytest tests/test_gpu_epnp.py -x -q -m gpu 2>&1 | tail -4
for g in 1 2 3 4; do
GROUP=$g DEPTHS=1,4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids
done
for f in 3 4 5; do
MR_EPNP_FIRST_ROUND=$f GROUP=2 DEPTHS=4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids
MR_EPNP_FIRST_ROUND=$f GROUP=4 DEPTHS=4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids
done
