#!/bin/bash
# Collect the round's profiling evidence on the GPU box (run through gpurun from the repo root).
#   tools/profile_round.sh r01
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary"
# 1) bench line (with CPU baseline)
python $GRAFT_REPO_ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
# 2) kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1
# 3) PMC passes, one group per run (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $OUT/pmc_lds -o p -- $BENCH > $OUT/pmc_lds.log 2>&1
find $OUT -name "*.csv" | head -40
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 > $OUT/rocminfo.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" > $OUT/lscpu.txt; nproc >> $OUT/lscpu.txt
