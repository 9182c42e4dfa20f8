#!/bin/bash
# Rehearsal of the N > 1 bench path on ONE GPU (gloo; the real run is RCCL, one GPU per rank).
export SVS_BENCH_BACKEND=gloo SVS_BENCH_VERIFY=1
mkdir -p gpurun_out
for n in 2 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n \
    bench.py --gpus $n --steps 40 --warmup 10 > gpurun_out/rehearse_$n.out 2> gpurun_out/rehearse_$n.err || { echo "n=$n FAILED"; tail -20 gpurun_out/rehearse_$n.err; exit 1; }
  grep '^{' gpurun_out/rehearse_$n.out | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['n_gpus'], round(d['value'],1), d['sharded_check'], d['config']['rows_per_gpu'])"
done
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
  bench.py --gpus 2 --steps 20 --warmup 4 --scaling weak --rows 300000 --dtype f16 > gpurun_out/rehearse_w.out 2> gpurun_out/rehearse_w.err || { echo "weak FAILED"; tail -20 gpurun_out/rehearse_w.err; exit 1; }
grep '^{' gpurun_out/rehearse_w.out | python -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['n_gpus'], round(d['value'],1), d['sharded_check'], d['config']['workload'])"
