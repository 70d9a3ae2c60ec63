#!/bin/bash
# The 1 / 2 / 4 / 8-GPU bench lines in one command (an 8-GPU MI355X node; each N launches its own ranks over RCCL, 127.0.0.1).
#   tools/scale.sh [steps] [warmup]        -> one JSON line per N on stdout (value = whole-job solves/s; efficiency is the reader's to compute)
# and, on stderr, one summary row per N: value, per-GPU value, the exchange's backend and the duration of one isolated collective
# (comm.us_per_collective: what a scaling curve is read against — at 88 KiB per rank the all-gather is latency, not bandwidth).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
STEPS=${1:-240}; WARMUP=${2:-24}
for n in 1 2 4 8; do
    out=$(python $ROOT/bench.py --gpus $n --steps $STEPS --warmup $WARMUP --no-cpu-baseline --no-secondary) || out="{\"n_gpus\": $n, \"error\": \"bench.py exited with $?\"}"
    echo "$out"
    echo "$out" | python -c "
import json, sys
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    c = d.get('comm') or {}
    print('N=%d  value %.3f M solves/s  per GPU %.3f M  ms/step %.4f  comm: %s  us_per_collective %s  algbw %s GB/s  rows verified %s' % (
        d['n_gpus'], d['value'] / 1e6, d['value'] / 1e6 / d['n_gpus'], d['ms_per_step'], (c.get('backend') or 'none (one rank)')[:60],
        c.get('us_per_collective'), c.get('algbw_GBps'), c.get('gathered_rows_verified')), file=sys.stderr)
except Exception as e:
    print('N=$n  no line (%r)' % (e,), file=sys.stderr)
"
done
