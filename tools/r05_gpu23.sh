for g in 1 3 4; do GROUP=$g DEPTHS=1,4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | sed 's/(streams found.*): / /'; done
