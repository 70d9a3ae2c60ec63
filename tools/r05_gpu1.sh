set -x
mkdir -p gpurun_out/r05
MR_PNP_SO=monorun_amd/variants/libmr_stamps.so python tools/gpu_hyp_timeline.py > gpurun_out/r05/hyp_timeline.txt 2>&1
DEPTHS=1,4,6,8 python tools/gpu_epnp_inflight.py > gpurun_out/r05/inflight_q4.txt 2>&1
GPU_MAX_HW_QUEUES=8 DEPTHS=4,6,8 python tools/gpu_epnp_inflight.py > gpurun_out/r05/inflight_q8.txt 2>&1
GPU_MAX_HW_QUEUES=16 DEPTHS=8,12 python tools/gpu_epnp_inflight.py > gpurun_out/r05/inflight_q16.txt 2>&1
MR_EPNP_FIRST_ROUND=4 DEPTHS=1,4 python tools/gpu_epnp_inflight.py > gpurun_out/r05/inflight_f4.txt 2>&1
MR_EPNP_FIRST_ROUND=2 DEPTHS=1,4 python tools/gpu_epnp_inflight.py > gpurun_out/r05/inflight_f2.txt 2>&1
cat gpurun_out/r05/*.txt
