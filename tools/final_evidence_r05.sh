#!/bin/bash
# The round's closing evidence on the final build (through gpurun, from the repo root): GPU tests, the profile set, the launch-set counters and trace,
# the long parity sweeps of the reference flow (default path) and of the fast mode, the fuzz runs, the spread of the bench line.
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/profile_r05.sh r05
bash tools/epnp_set_valu.sh > gpurun_out/r05_epnp_valu_per_launch_set.txt 2>&1
bash tools/profile_epnp_inflight.sh 4 > gpurun_out/r05_inflight_trace.txt 2>&1
TAG=r05 NSEEDS=${NSEEDS:-1000} TRIALS=${TRIALS:-2000} REPEATS=8 PARTS="epnp fuzz_epnp k0 fuzz repeats" bash tools/gpu_long_evidence.sh
