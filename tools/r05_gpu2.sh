mkdir -p gpurun_out/r05
for q in 4 8 16; do
for f in 2 3 8; do
GPU_MAX_HW_QUEUES=$q MR_EPNP_FIRST_ROUND=$f DEPTHS=4,8,12 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | sed "s/^/Q$q /" >> gpurun_out/r05/inflight_matrix.txt
done; done
cat gpurun_out/r05/inflight_matrix.txt
