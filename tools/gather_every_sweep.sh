#!/bin/bash
# steps per exchange of the sharded pipeline, through the N > 1 code path with the one rank a 1-GPU box has
# (SVS_BENCH_FORCE_DIST=1), on one GPU's share of the 8-GPU strong-scaling run.  usage: tools/gather_every_sweep.sh [rows=125000]
rows=${1:-125000}
cd "$(dirname "$0")/.."
export SVS_BENCH_FORCE_DIST=1
for rep in 1 2; do
for ge in 4 8 16 32; do
  for inflight in 2 3; do
  timeout -k 10 120 python bench.py --rows $rows --steps 800 --warmup 64 --gather-every $ge --inflight $inflight --no-cpu-baseline --batch "" --concurrent 0 --configs "" --kb 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('rows %8d  steps per exchange %2d  in flight %d  %8.1f queries/s  %.4f ms per step' % ($rows, $ge, $inflight, d['value'], d['ms_per_step']))
"
  done
done
done
