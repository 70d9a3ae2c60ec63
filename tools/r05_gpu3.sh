mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_epnp.py -x -q -m gpu 2>&1 | tail -5
MR_PNP_SO=monorun_amd/variants/libmr_stamps.so python tools/gpu_hyp_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/hyp_timeline_new.txt
DEPTHS=1,4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/inflight_new_q3.txt
MR_PNP_SO=monorun_amd/variants/libmr_q2.so DEPTHS=1,4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/inflight_new_q2.txt
bash tools/profile_epnp_quick.sh 2>&1 | tail -12
