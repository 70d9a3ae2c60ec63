for f in 3 4 6 8; do
MR_EPNP_FIRST_ROUND=$f GROUP=3 DEPTHS=4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids
done
MR_PNP_SO=monorun_amd/variants/libmr_q2.so GROUP=3 DEPTHS=4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids
MR_PNP_SO=monorun_amd/variants/libmr_q2.so MR_EPNP_FIRST_ROUND=4 GROUP=3 DEPTHS=4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids
