#!/bin/bash
# The round's closing evidence on the final build (through gpurun, from the repo root): GPU tests, the profile set of the reference flow (PMC passes first, so that
# the bench lines replay this round's counters), the launch-set counters and the in-flight trace, the million-object parity sweeps of the default flow and of the fast
# mode, the 2 000-trial fuzz runs, the spread of the bench line.   bash tools/final_evidence.sh [tag]     (then, here: python tools/summarize_flow.py <tag>)
TAG=${1:-r06}
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/${TAG}_gpu_tests.txt
bash tools/profile_flow.sh $TAG
bash tools/epnp_set_valu.sh > gpurun_out/${TAG}_epnp_valu_per_launch_set.txt 2>&1
bash tools/profile_epnp_inflight.sh 4 > gpurun_out/${TAG}_inflight_trace.txt 2>&1
(for w in 0 rule; do if [ $w = rule ]; then NBATCH=12 python tools/gpu_wide_ab.py; else MR_EP_WIDE=$w NBATCH=12 python tools/gpu_wide_ab.py; fi; done) 2>&1 | grep objects > gpurun_out/${TAG}_wide_vs_quad.txt
TAG=$TAG NSEEDS=${NSEEDS:-1000} TRIALS=${TRIALS:-2000} REPEATS=8 PARTS="${PARTS:-epnp fuzz_epnp k0 fuzz repeats}" bash tools/gpu_long_evidence.sh
