#!/bin/bash
# Does the order of the bench's untimed work change the driver-command window?  N processes each way.
# usage: tools/gpu_bench_order.sh [N]   -> gpurun_out/bench_order.txt
N=${1:-6}
mkdir -p gpurun_out
out=gpurun_out/bench_order.txt
: > $out
for k in $(seq $N); do
  for m in 0 1; do
    v=$(MR_BENCH_DIAG_FIRST=$m MR_BENCH_DEBUG=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2> gpurun_out/bench_order.err | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "diag_first=$m run $k value $v $(grep -m1 'steps issued' gpurun_out/bench_order.err)" >> $out
  done
done
cat $out
