#!/bin/bash
# Round-5 profiling evidence (run through gpurun from the repo root): the default bench line (reference flow, launch sets in flight), its
# rocprofv3 kernel trace, the one-call-at-a-time trace of the flow, PMC passes (instructions, HBM traffic) in their own runs, the K0 fast
# mode's line and trace, the stress line.   bash tools/profile_r05.sh [tag]
TAG=${1:-r05}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python $R/bench.py --steps 240 --warmup 24 --no-cpu-baseline --no-secondary > $OUT/bench_240.json 2>> $OUT/bench.err
python $R/bench.py --workload stress > $OUT/bench_stress.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_bench -o t -- python $R/bench.py --steps 96 --warmup 12 --no-cpu-baseline --no-secondary > $OUT/trace_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_k0 -o t -- python $R/bench.py --flow k0 --steps 96 --warmup 12 --no-cpu-baseline --no-secondary > $OUT/trace_k0.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_call -o t -- env REPS=10 python $R/tools/gpu_epnp_path.py > $OUT/trace_call.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- env REPS=6 python $R/tools/gpu_epnp_path.py > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- env REPS=6 python $R/tools/gpu_epnp_path.py > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_lds -o p -- env REPS=6 python $R/tools/gpu_epnp_path.py > $OUT/pmc_lds.log 2>&1
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 > $OUT/rocminfo.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > $OUT/lscpu.txt; nproc >> $OUT/lscpu.txt
find $OUT -name "*.csv" | wc -l
