#!/bin/bash
# What ONE GPU of the strong-scaling configs[1] run (1M x 1536 f32 over N GPUs) does per step, measured on one GPU:
# bench.py over 1M / N rows, alone (no collective) and through the N > 1 code path with the one rank a 1-GPU box has
# (SVS_BENCH_FORCE_DIST=1: RCCL all-gather per exchange + host merge).  8 x value(125k rows) vs value(1M rows) is the
# ceiling of the 8-GPU line; the forced-distributed column adds the exchange's fixed cost.
# usage: tools/strong_projection.sh [steps=400]
steps=${1:-400}
cd "$(dirname "$0")/.."
for rows in 1000000 500000 250000 125000; do
  for fd in 0 1; do
    if [ $fd = 1 ]; then export SVS_BENCH_FORCE_DIST=1; else unset SVS_BENCH_FORCE_DIST; fi
    timeout -k 10 120 python bench.py --rows $rows --steps $steps --warmup 40 --no-cpu-baseline --batch "" --concurrent 0 --configs "" --kb 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('rows %8d  forced-distributed %d  %8.1f queries/s  %.4f ms per step  score %.4f ms  select %.4f ms  in flight %s  steps per exchange %s' % (
            $rows, $fd, d['value'], d['ms_per_step'], d['stage_ms']['score'], d['stage_ms']['select'], d['config'].get('searches_in_flight'), d['config'].get('steps_per_exchange')))
"
  done
done
