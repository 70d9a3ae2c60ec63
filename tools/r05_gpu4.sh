python -m pytest tests/test_gpu_epnp.py tests/test_coord2d.py tests/test_roi_align.py -x -q -m gpu 2>&1 | tail -4
for so in monorun_amd/libmonorun_pnp.so monorun_amd/variants/libmr_q2.so; do
for o in 1024 2048 4096; do
MR_PNP_SO=$so OBJECTS=$o DEPTHS=1,4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids
done; done
MR_EPNP_FIRST_ROUND=3 OBJECTS=2048 DEPTHS=4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids
MR_EPNP_FIRST_ROUND=3 OBJECTS=4096 DEPTHS=4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids
