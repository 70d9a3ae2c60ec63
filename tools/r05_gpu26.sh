for g in 5 8; do for f in 2 3 4; do MR_EPNP_FIRST_ROUND=$f NBATCH=$((g*4)) GROUP=$g DEPTHS=4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | sed 's/(streams found.*): / /'; done; done
NBATCH=24 GROUP=8 DEPTHS=3 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | sed 's/(streams found.*): / /'
NBATCH=16 GROUP=8 DEPTHS=2 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | sed 's/(streams found.*): / /'
NBATCH=16 GROUP=4 DEPTHS=4 python tools/gpu_epnp_inflight.py 2>&1 | grep -v amdgpu.ids | sed 's/(streams found.*): / /'
