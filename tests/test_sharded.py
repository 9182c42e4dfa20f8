"""Row sharding.  CPU: the gather + host-merge logic with
world_size-2 gloo processes and a test-double local search; the result must equal
the unsharded oracle for every G.  GPU: two shards on one card vs one index."""
import os
import socket
import sys

import numpy as np
import pytest

from oracle import svs_oracle as oracle
from synth import corpus_and_query

from svs_amd.sharded import (ShardedIndex, merge_topk, merge_topk_batch, pack_record, record_layout, shard_bounds,
                             unpack_records)

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_bounds_cover_rows_once():
    for n in (0, 1, 7, 8, 9, 1000, 1_000_000, 100_000_001):
        for g in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, g, r) for r in range(g)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b and c <= d


def test_merge_is_the_total_order():
    rng = np.random.default_rng(5)
    v = rng.standard_normal(5000).astype(np.float32)
    v[rng.choice(5000, 800)] = 0.5          # ties across shards
    v[10] = 0.0; v[4000] = -0.0
    k = 300
    for g in (1, 2, 4, 8):
        ss, rr = [], []
        for r in range(g):
            lo, hi = shard_bounds(5000, g, r)
            top = oracle.total_order_top_k(v[lo:hi], k)
            s = np.full(k, -np.inf, np.float32); rw = np.full(k, -1, np.int64)
            s[:len(top)] = [a for a, _ in top]; rw[:len(top)] = [b + lo for _, b in top]
            ss.append(s); rr.append(rw)
        ms, mr = merge_topk(np.stack(ss), np.stack(rr), k)
        exp = oracle.total_order_top_k(v, k)
        assert [int(x) for x in mr] == [i for _, i in exp]
        assert [float(x) for x in ms] == [s for s, _ in exp]


def test_record_wire_format_roundtrip():
    """The byte record bench.py exchanges with one all-gather per query."""
    rng = np.random.default_rng(2)
    for k in (1, 3, 100, 101):
        s_off, rec = record_layout(k)
        assert s_off % 8 == 0 and rec == s_off + 8 * k
        world = 4
        recs, exp_s, exp_r = [], [], []
        for r in range(world):
            c = k if r != 2 else max(k - 2, 0)          # one shard holds fewer than k rows
            sc = np.sort(rng.standard_normal(c).astype(np.float32))[::-1]
            rw = rng.integers(0, 1 << 40, c)
            recs.append(pack_record(sc, rw, k)); exp_s.append(sc); exp_r.append(rw)
        sc, rw = unpack_records(np.stack(recs), world, k)
        for r in range(world):
            c = len(exp_s[r])
            assert np.array_equal(sc[r, :c], exp_s[r]) and np.array_equal(rw[r, :c], exp_r[r])
            assert np.all(rw[r, c:] == -1) and np.all(np.isneginf(sc[r, c:]))
        ms, mr = merge_topk(sc, rw, k)
        assert len(ms) == k and np.all(np.diff(ms) <= 0) and np.all(mr >= 0)


def test_batched_merge_equals_per_query_merge():
    """merge_topk_batch (one lexsort over (nq, G*k)) == merge_topk query by query, with ties,
    signed zeros, padding and a NaN (MultiDeviceIndex / ShardedIndex use the batched form)."""
    rng = np.random.default_rng(1)
    g, nq, k = 4, 9, 20
    sc = rng.standard_normal((g, nq, k)).astype(np.float32)
    rw = rng.integers(0, 10 ** 6, (g, nq, k))
    sc[0, :, 5] = 0.0; sc[1, :, 3] = -0.0; sc[2, :, 7:] = -np.inf; rw[2, :, 7:] = -1
    sc[:, :, 11] = 0.25                       # a tie across every shard
    sc[3, 2, 1] = np.nan
    bs, br = merge_topk_batch(sc, rw, 15)
    for i in range(nq):
        ms, mr = merge_topk(sc[:, i, :], rw[:, i, :], 15)
        assert np.array_equal(mr, br[i]) and np.array_equal(ms, bs[i], equal_nan=True)


def test_batched_merge_on_keys_and_on_lexsort_agree():
    """merge_topk_batch orders rows below 2^32 by ONE sort of 64-bit keys (the kernels' own order key, csrc/keys.h) and
    anything larger by a two-key lexsort: both must give merge_topk's answer on zeros of both signs, infinities,
    denormals, NaN, ties across shards and padding -- 300 random tables each way."""
    rng = np.random.default_rng(12)
    vals = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, 0.5, np.nan, 2e-38, -3e-39, 0.25, 0.25], dtype=np.float32)
    for trial in range(300):
        g, nq, kk = int(rng.integers(1, 6)), int(rng.integers(1, 5)), int(rng.integers(1, 12))
        sc = rng.choice(vals, size=(g, nq, kk))
        rw = np.stack([rng.permutation(1000)[:nq * kk].reshape(nq, kk) + 1000 * i for i in range(g)]).astype(np.int64)
        if trial % 2:
            rw += (1 << 32)                   # the lexsort branch
        pad = rng.random((g, nq, kk)) < 0.2
        sc = np.where(pad, -np.inf, sc).astype(np.float32)
        rw = np.where(pad, -1, rw)
        k = int(min((rw >= 0).sum(axis=(0, 2)).min(), rng.integers(1, 20)))
        if k == 0:
            continue
        bs, br = merge_topk_batch(sc, rw, k)
        for i in range(nq):
            ms, mr = merge_topk(sc[:, i, :], rw[:, i, :], k)
            assert np.array_equal(mr, br[i]), (trial, i)
            assert np.array_equal(ms + np.float32(0.0), bs[i] + np.float32(0.0), equal_nan=True), (trial, i)


def test_record_of_a_query_batch():
    """A record holds the result of ONE call of nq queries: [nq*k scores | pad | nq*k rows]."""
    k, nq, world = 7, 5, 3
    s_off, rec = record_layout(k, nq)
    assert s_off % 8 == 0 and s_off >= nq * k * 4 and rec == s_off + nq * k * 8
    rng = np.random.default_rng(3)
    recs, ss, rr = [], [], []
    for r in range(world):
        sc = -np.sort(-rng.standard_normal((nq, k - r)).astype(np.float32), axis=1)
        rw = rng.integers(0, 1 << 40, (nq, k - r))
        recs.append(pack_record(sc, rw, k, nq)); ss.append(sc); rr.append(rw)
    sc, rw = unpack_records(np.stack(recs), world, k, nq)
    sc, rw = sc.reshape(world, nq, k), rw.reshape(world, nq, k)
    for r in range(world):
        assert np.array_equal(sc[r, :, : k - r], ss[r]) and np.array_equal(rw[r, :, : k - r], rr[r])
        assert np.all(rw[r, :, k - r:] == -1)


def _bench_like_worker(rank, world, port, out_q):
    """bench.py's exchange, on gloo/CPU tensors: byte records, all_gather_into_tensor
    with async_op, then unpack + merge on rank 0."""
    sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, d, k, steps = 3001, 32, 20, 5
    m, qs = corpus_and_query("gaussian", 99, n, d, steps)
    lo, hi = shard_bounds(n, world, rank)
    _, rec = record_layout(k)
    local = torch.zeros((steps, rec), dtype=torch.uint8)
    gathered = torch.zeros((steps, world, rec), dtype=torch.uint8)
    works = []
    for i in range(steps):
        top = oracle.total_order_top_k(oracle.cpu_scores(m[lo:hi], qs[i]), k)
        local[i] = torch.from_numpy(pack_record(np.array([s for s, _ in top], np.float32),
                                                np.array([r + lo for _, r in top], np.int64), k))
        works.append(dist.all_gather_into_tensor(gathered[i].view(-1), local[i], async_op=True))
    for w in works:
        w.wait()
    if rank == 0:
        res = []
        for i in range(steps):
            sc, rw = unpack_records(gathered[i].numpy(), world, k)
            ms, mr = merge_topk(sc, rw, k)
            res.append((ms.tolist(), mr.tolist()))
        out_q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_exchange_on_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_like_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m, qs = corpus_and_query("gaussian", 99, 3001, 32, 5)
    for i, (ms, mr) in enumerate(res):
        exp = oracle.total_order_top_k(oracle.cpu_scores(m, qs[i]), 20)
        assert mr == [r for _, r in exp]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, d, k, seed, out_q):
    sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
    import torch.distributed as dist
    from fake_backend import OracleIndex
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, qs = corpus_and_query("gaussian", seed, n, d, 3)
    lo, hi = shard_bounds(n, world, rank)
    local = OracleIndex(m[lo:hi], row_offset=lo)
    sh = ShardedIndex(local.search_batch, n_total=n)
    res = sh.search_batch(qs, k)
    one = sh.search(qs[0], k)
    if rank == 0:
        out_q.put((res[0].tolist(), res[1].tolist(), one))
    else:
        assert res is None and one is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,k", [(5003, 100), (64, 100), (3, 5)])
def test_gloo_world2_matches_unsharded(n, k):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    d, seed = 48, 31
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, d, k, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    scores, rows, one = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m, qs = corpus_and_query("gaussian", seed, n, d, 3)
    for qi, qv in enumerate(qs):
        exp = oracle.total_order_top_k(oracle.cpu_scores(m, qv), k)
        assert rows[qi] == [i for _, i in exp]
        assert np.allclose(scores[qi], [s for s, _ in exp], atol=1e-6)
    assert [i for _, i in one] == rows[0]


@pytest.mark.gpu
def test_two_shards_on_one_gpu_equal_one_index(gpu):
    """Same kernel, same summation order wherever a row lives: the merged result
    of any sharding is IDENTICAL (bitwise scores) to the single-index result."""
    from svs_amd import DeviceIndex
    n, d, k = 70001, 1536, 100
    m, qs = corpus_and_query("gaussian", 41, n, d, 4)
    whole = DeviceIndex(m)
    for g in (2, 3, 8):
        shards = []
        for r in range(g):
            lo, hi = shard_bounds(n, g, r)
            shards.append(DeviceIndex(m[lo:hi], row_offset=lo))
        ws, wr = whole.search_batch(qs, k)
        parts = [s.search_batch(qs, k) for s in shards]
        for qi in range(len(qs)):
            ms, mr = merge_topk(np.stack([p[0][qi] for p in parts]), np.stack([p[1][qi] for p in parts]), k)
            assert np.array_equal(mr, wr[qi]) and np.array_equal(ms, ws[qi])
        for s in shards:
            s.release()
    whole.release()


@pytest.mark.gpu
def test_multi_device_index_in_one_process(gpu):
    """MultiDeviceIndex (threads, no RCCL): shards on the same card here; results must be
    identical to one index, through searches, appends and tombstones, and under the KB."""
    import functools
    import svs_amd
    from svs_amd import DeviceIndex, MultiDeviceIndex
    n, d, k = 50003, 768, 100
    m, qs = corpus_and_query("gaussian", 43, n, d, 5)
    whole = DeviceIndex(m[:40000])
    multi = MultiDeviceIndex(m[:40000], devices=[0, 0, 0])
    assert multi.shape == (40000, d) and multi.devices == [0, 0, 0]
    for q in qs:
        assert multi.search(q, k) == whole.search(q, k)
    ws, wr = whole.search_batch(qs, 33)
    ms, mr = multi.search_batch(qs, 33)
    assert np.array_equal(ws, ms) and np.array_equal(wr, mr)
    whole.append(m[40000:]); multi.append(m[40000:])
    dead = [5, 13334, 26667, 39999, 40000, 50002] + [r for _, r in whole.search(qs[1], 10)]
    whole.mask_rows(dead); multi.mask_rows(dead)
    assert multi.n_masked == whole.n_masked == len(set(dead))
    for q in qs:
        assert multi.search(q, k) == whole.search(q, k)
    assert len(multi.search(qs[0], 10 ** 6)) == n - len(set(dead))
    with pytest.raises(ValueError):
        multi.search(qs[0][:5], 3)
    held = multi.share()
    multi.release()
    assert held.search(qs[2], 7) == whole.search(qs[2], 7)      # a shared owner keeps every shard alive
    held.release(); whole.release()
    tiny = MultiDeviceIndex(m[:2], devices=[0, 0, 0, 0])          # fewer rows than shards
    assert [r for _, r in tiny.search(qs[0], 5)] == [r for _, r in oracle.total_order_top_k(oracle.cpu_scores(m[:2], qs[0]), 5)]
    tiny.release()


@pytest.mark.gpu
def test_pipelined_exchange_world1_equals_search(gpu):
    """ShardedIndex's pipelined path (what bench.py times) at one rank: records written by the
    search kernel straight into pinned host memory, alternating streams -- identical to the blocking
    search, query by query; and the blocking record path (search_batch) likewise."""
    import torch
    from svs_amd import DeviceIndex
    n, d, k = 60000, 1536, 100
    m, qs = corpus_and_query("gaussian", 47, n, d, 21)
    idx = DeviceIndex(m, row_offset=5_000_000)
    dev = torch.device("cuda:0")
    sh = ShardedIndex(idx, n_total=n, device=dev, gather_every=8, streams=2)
    qt = torch.from_numpy(qs).to(dev)
    sh.open(len(qs), k)
    for i in range(len(qs)):
        assert sh.enqueue(qt[i].data_ptr(), d) == i
    res = sh.collect()
    for i, (s, r) in enumerate(res):
        one = idx.search(qs[i], k)
        assert [int(x) for x in r] == [x for _, x in one] and [float(x) for x in s] == [x for x, _ in one]
    bs, br = sh.search_batch(qs, k)
    ws, wr = idx.search_batch(qs, k)
    assert np.array_equal(bs, ws) and np.array_equal(br, wr)
    assert sh.search(qs[3], 7) == idx.search(qs[3], 7)
    idx.release()


def _nccl_world1_worker(port, out_q):
    """Child process: RCCL itself (backend "nccl") with one rank on cuda:0 -- process-group init with a
    device id, all_gather_into_tensor on HBM records (blocking and async_op), the host merge."""
    import os
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        import torch
        import torch.distributed as dist
        from svs_amd import DeviceIndex
        from svs_amd.sharded import ShardedIndex
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        n, d, k = 50000, 1536, 100
        m, qs = corpus_and_query("gaussian", 53, n, d, 19)
        idx = DeviceIndex(m, row_offset=123_000)
        # (force_collective: one rank would otherwise skip the exchange altogether)
        sh = ShardedIndex(idx, n_total=n, device=dev, gather_every=8, streams=2, force_collective=True)
        assert sh.gather_every == 8 and sh.streams >= 2
        qt = torch.from_numpy(qs).to(dev)
        sh.open(len(qs), k)
        for i in range(len(qs)):
            sh.enqueue(qt[i].data_ptr(), d)
        res = sh.collect()
        ok = True
        for i, (s, r) in enumerate(res):
            one = idx.search(qs[i], k)
            ok &= [int(x) for x in r] == [x for _, x in one] and [float(x) for x in s] == [x for x, _ in one]
        bs, br = sh.search_batch(qs, k)
        ws, wr = idx.search_batch(qs, k)
        ok &= bool(np.array_equal(bs, ws) and np.array_equal(br, wr))
        idx.release()
        dist.destroy_process_group()
        out_q.put("ok" if ok else "results differ")
    except BaseException as e:   # noqa: BLE001
        out_q.put(f"{type(e).__name__}: {e}")


@pytest.mark.gpu
def test_pipelined_exchange_over_rccl_world1(gpu):
    """The exchange on the REAL backend: the multi-GPU tests above run gloo (CPU tensors allowed, no device
    binding); this one initialises RCCL with a device id and gathers CUDA tensors, on the one rank the box has.
    In a child process: a process group is per process, and RCCL should not live in the test runner."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(29533, q))
    p.start()
    p.join(timeout=600)
    assert not p.is_alive(), "RCCL worker hung"
    assert q.get(timeout=5) == "ok"
