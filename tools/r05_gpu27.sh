python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/profile_r05.sh r05
bash tools/epnp_set_valu.sh > gpurun_out/r05_epnp_valu_per_launch_set.txt 2>&1
bash tools/profile_epnp_inflight.sh 4 > gpurun_out/r05_inflight_trace.txt 2>&1
