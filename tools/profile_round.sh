#!/bin/bash
# Collect the round's profiling evidence on the GPU box (run through gpurun from the repo root).
#   tools/profile_round.sh r03
# Two regimes of the config-2 bench are profiled: ONE stream (every launch isolated: the 4-waves-per-object kernel, what
# `roofline.achieved` is computed from) and 4 launches IN FLIGHT (the headline: the 2-waves-per-object kernel); plus the NOC path
# (K2 decode, fused head->pose) and the EPnP/RANSAC initialiser.  Counters in their own --pmc passes (kernel trace only).
set -u
TAG=${1:-r04}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PARTS=${PARTS:-bench stream1 inflight noc epnp}
# 1) the bench line (with CPU baselines and secondary figures), exactly as the driver runs it, and the long window
if [[ " $PARTS " == *" bench "* ]]; then
python $R/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python $R/bench.py --steps 240 --warmup 24 --no-cpu-baseline --no-secondary > $OUT/bench_240.json 2>> $OUT/bench.err
fi
prof() {   # prof <subdir> <counters or ""> -- <command...>
    local sub=$1 pmc=$2; shift 3
    if [ -z "$pmc" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$sub -o t -- "$@" > $OUT/$sub.log 2>&1
    else rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $OUT/$sub -o p -- "$@" > $OUT/$sub.log 2>&1; fi
}
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE GRBM_COUNT"
if [[ " $PARTS " == *" stream1 "* ]]; then
    B="env MR_BENCH_PREWARM_LAUNCHES=0 python $R/bench.py --steps 96 --warmup 12 --no-cpu-baseline --no-secondary --in-flight 1"
    prof trace1 "" -- $B
    prof pmc1_fetch "FETCH_SIZE" -- $B
    prof pmc1_write "WRITE_SIZE" -- $B
    prof pmc1_sq "$SQ1" -- $B
    prof pmc1_lds "$SQ2" -- $B
fi
if [[ " $PARTS " == *" inflight "* ]]; then
    # kernel trace of the headline regime (4 launches in flight: overlapping durations of the 2-waves-per-object kernel) ...
    prof trace4 "" -- python $R/bench.py --steps 96 --warmup 12 --no-cpu-baseline --no-secondary --in-flight 4
    # ... and its counters from isolated launches of the SAME kernel (--waves 2 on one stream): counter collection serialises
    # kernels anyway (under it the pipeline's overlap self-test finds no two streams that run side by side and falls back to depth 1)
    B="env MR_BENCH_PREWARM_LAUNCHES=0 python $R/bench.py --steps 96 --warmup 12 --no-cpu-baseline --no-secondary --in-flight 1 --waves 2"
    prof pmc4_fetch "FETCH_SIZE" -- $B
    prof pmc4_write "WRITE_SIZE" -- $B
    prof pmc4_sq "$SQ1" -- $B
    prof pmc4_lds "$SQ2" -- $B
fi
# 2) the NOC path at B = 1024 (3 distinct resident head outputs, 276 MiB)
if [[ " $PARTS " == *" noc "* ]]; then
for W in k2 fused; do
    prof noc_${W}_trace "" -- env WHICH=$W REPS=60 python $R/tools/gpu_noc_path.py
    prof noc_${W}_fetch "FETCH_SIZE" -- env WHICH=$W REPS=60 python $R/tools/gpu_noc_path.py
    prof noc_${W}_write "WRITE_SIZE" -- env WHICH=$W REPS=60 python $R/tools/gpu_noc_path.py
    prof noc_${W}_sq "$SQ1" -- env WHICH=$W REPS=60 python $R/tools/gpu_noc_path.py
done
fi
# 3) the EPnP / RANSAC initialiser
if [[ " $PARTS " == *" epnp "* ]]; then
prof epnp_trace "" -- env REPS=10 python $R/tools/gpu_epnp_path.py
prof epnp_sq "$SQ1" -- env REPS=6 python $R/tools/gpu_epnp_path.py
prof epnp_lds "$SQ2" -- env REPS=6 python $R/tools/gpu_epnp_path.py
prof epnp_fetch "FETCH_SIZE" -- env REPS=6 python $R/tools/gpu_epnp_path.py
prof epnp_write "WRITE_SIZE" -- env REPS=6 python $R/tools/gpu_epnp_path.py
fi
find $OUT -name "*.csv" | wc -l
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 > $OUT/rocminfo.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > $OUT/lscpu.txt; nproc >> $OUT/lscpu.txt
