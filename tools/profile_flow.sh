#!/bin/bash
# The round's profiling evidence for the reference flow as the bench's `value` (run through gpurun from the repo root):
#   1. PMC passes (HBM traffic, instruction counts) of one call at a time, each in its own run, and the one-call-at-a-time kernel trace;
#      the same traffic passes on the stress shape;
#   2. tools/summarize_flow.py condenses them into profiles/<tag>_* ON THE BOX, so that
#   3. the bench lines (default, 240 steps, stress) that follow replay THIS round's `traffic` / `valu_issue` (the bench copy
#      used to predate the traffic pass);
#   4. kernel traces of the bench regimes (launch sets in flight; --flow k0).
# Everything lands in gpurun_out/prof_<tag>; `python tools/summarize_flow.py <tag>` here condenses it into the tracked profiles/<tag>_* files.
#   bash tools/profile_flow.sh [tag]
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_call -o t -- env REPS=10 python $R/tools/gpu_epnp_path.py > $OUT/trace_call.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_call_B100 -o t -- env REPS=10 OBJECTS=100 python $R/tools/gpu_epnp_path.py > $OUT/trace_call_B100.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- env REPS=6 python $R/tools/gpu_epnp_path.py > $OUT/pmc_$c.log 2>&1
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_stress_$c -o p -- env REPS=4 STRESS=1 python $R/tools/gpu_epnp_path.py > $OUT/pmc_stress_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- env REPS=6 python $R/tools/gpu_epnp_path.py > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds -o p -- env REPS=6 python $R/tools/gpu_epnp_path.py > $OUT/pmc_lds.log 2>&1
python $R/tools/summarize_flow.py $TAG > $OUT/summarize_on_box.log 2>&1        # profiles/<tag>_epnp_traffic*.json, _epnp_valu_per_launch.json exist from here on
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python $R/bench.py --steps 240 --warmup 24 --no-cpu-baseline --no-secondary > $OUT/bench_240.json 2>> $OUT/bench.err
python $R/bench.py --workload stress > $OUT/bench_stress.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -o t -- python $R/bench.py --steps 96 --warmup 12 --no-cpu-baseline --no-secondary > $OUT/trace_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k0 -o t -- python $R/bench.py --flow k0 --steps 96 --warmup 12 --no-cpu-baseline --no-secondary > $OUT/trace_k0.log 2>&1
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 > $OUT/rocminfo.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > $OUT/lscpu.txt; nproc >> $OUT/lscpu.txt
find $OUT -name "*.csv" | wc -l
