#!/bin/bash
# The 1 / 2 / 4 / 8-GPU bench lines in one command (an 8-GPU MI355X node; each N launches its own ranks over RCCL, 127.0.0.1).
#   tools/scale.sh [steps] [warmup]        -> one JSON line per N on stdout (value = whole-job solves/s; efficiency is the reader's to compute)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
STEPS=${1:-240}; WARMUP=${2:-24}
for n in 1 2 4 8; do
    python $ROOT/bench.py --gpus $n --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-secondary || echo "{\"n_gpus\": $n, \"error\": \"bench.py exited with $?\"}"
done
