"""Child process of tests/test_sharded_gpu.py: ONE rank of a row-sharded search that
drives the library's pipelined device exchange -- ShardedIndex.open / enqueue / collect, the path
bench.py --gpus N times -- over gloo, the ranks sharing the one card of the test box (RCCL needs a
GPU per rank).  Started as a fresh interpreter BEFORE anything touches the GPU.
  usage: sharded_gpu_worker.py RANK WORLD PORT OUT.npz N D NQ K GATHER_EVERY DTYPE"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out = sys.argv[4]
    n, d, nq, k, ge = (int(x) for x in sys.argv[5:10])
    dtype = sys.argv[10]
    import numpy as np
    import torch
    import torch.distributed as dist
    from svs_amd import DeviceIndex
    from svs_amd.sharded import ShardedIndex, shard_bounds
    from synth import corpus_and_query

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    m, qs = corpus_and_query("gaussian", 2024, n, d, nq)
    lo, hi = shard_bounds(n, world, rank)
    idx = DeviceIndex(m[lo:hi], device=0, dtype=dtype, row_offset=lo)
    sh = ShardedIndex(idx, n, device=dev, gather_every=ge, streams=2)
    assert sh.gather_every == ge and sh.streams >= 2
    qd = torch.from_numpy(qs).to(dev)
    torch.cuda.synchronize()
    sh.open(nq, k)
    for i in range(nq):
        sh.enqueue(qd[i].data_ptr(), d)
    res = sh.collect()
    if rank == 0:
        assert res is not None and len(res) == nq
        np.savez(out, scores=np.stack([s for s, _ in res]), rows=np.stack([r for _, r in res]))
    else:
        assert res is None
    dist.barrier()
    idx.release()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
