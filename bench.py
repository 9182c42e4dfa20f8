#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the svs_amd hot path on MI355X.

Metric (BASELINE.json): queries/sec + p50 latency, cosine top-100 over
1M x 1536 fp32.  One "step" = one pass of the hot path over one batch of
synthetic input = ONE single-query search (score stage + top-k) over the whole
corpus, inputs already resident in HBM.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

N > 1: the 1M-row corpus is row-sharded; every rank scores its
shard for every query, local top-k lists are exchanged with one RCCL all-gather
per query and merged on rank 0 (strong scaling: total work is fixed).
`--scaling weak` instead keeps --rows rows PER GPU.

Prints ONE JSON line on rank 0 (see the driver contract in the task notes):
value = whole-job queries/s; roofline = algorithmic corpus bytes per launch of
the dominant kernel / its HIP-event-measured duration; cpu_baseline = the numpy
restatement of the reference path (oracle/) timed on this host's cores.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md, chip-level parameters)
MFMA_PEAK = {"f16": 2.5e15, "fp8": 5.0e15}  # dense FLOP/s (same table; never the 2:1-sparsity figures)
KERNEL_NAME = {  # dominant kernel per dtype and batch shape (svs_amd/csrc)
    ("f32", 1): "gemv_f32_oneshot_kernel (gemv_f32.h)", ("f16", 1): "gemv_f16_oneshot_kernel (gemv_f16.h)",
    ("fp8", 1): "gemv_fp8_oneshot_kernel (fp8.h)", ("f32", 16): "gemm_q16r_kernel<FUSE, 4> (gemm_q16.h)",
    ("f32", 256): "gemm_tiled_kernel<64, FUSE, 4> (gemm_tiled.h)", ("f16", 1024): "gemm_phased_kernel<FUSE, 2> (gemm_phased.h)",
    ("fp8", 256): "gemm_phased_kernel<FUSE, 1> (gemm_phased.h)",
}


def batched_config(torch, idx, name, n, d, dtype, nq, k, seed, reps):
    """Secondary figure: one BASELINE.json batch configuration on an index already in HBM.  Kernel
    time = HIP events around the dominant kernel launch inside the library (svs_timing_t.dominant_ms_sum)."""
    dev = torch.device("cuda", idx.device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    q = torch.randn((nq, d), device=dev, dtype=torch.float32, generator=g)
    q = (q / q.norm(dim=1, keepdim=True)).cpu().numpy()
    for _ in range(3):
        idx.search_batch(q, k)
    idx.set_timing(True)
    a = time.perf_counter()
    for _ in range(reps):
        idx.search_batch(q, k)
    dt = time.perf_counter() - a
    score_ms, select_ms, cnt = idx.get_timing()
    dom_ms = idx.last_dominant_ms_sum / max(cnt, 1)
    idx.set_timing(False)
    esz = {"f32": 4, "f16": 2, "fp8": 1}[dtype]
    flops = 2.0 * n * d * nq
    bytes_ = float(n) * d * esz + (4.0 * n if dtype == "fp8" else 0.0)
    pf = flops / (dom_ms * 1e-3)
    out = {"workload": name, "queries_per_call": nq, "ms_per_call": dt / reps * 1e3, "value": nq * reps / dt, "unit": "queries/s",
           "stage_ms": {"score": score_ms / max(cnt, 1), "select": select_ms / max(cnt, 1), "dominant_kernel": dom_ms},
           "roofline": {"bound": "mfma", "kernel": KERNEL_NAME.get((dtype, nq), "batched GEMM"), "achieved": pf / 1e12,
                        "peak": MFMA_PEAK[dtype] / 1e12, "unit": "TFLOP/s", "frac": pf / MFMA_PEAK[dtype],
                        "algorithmic_flops_per_launch": flops, "avg_launch_ms": dom_ms, "launches_timed": cnt,
                        "traffic": None},
           "note": "host API (queries in, results out, synchronised); kernel time from HIP events inside the library"}
    # HBM bytes of that kernel from the committed FETCH_SIZE pass (tools/profile_round.sh), if there is one
    cfg_tag = {("f16", 1024): "cfg2", ("fp8", 256): "cfg4"}.get((dtype, nq))
    if cfg_tag:
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{cfg_tag}_summary.json")), reverse=True):
            try:
                pm = json.load(open(f)).get("pmc", {})
                hit = [v["hbm_read_bytes_per_launch"] for kn, v in pm.items() if "gemm_phased_kernel<true" in kn and "hbm_read_bytes_per_launch" in v]
                if hit:
                    out["roofline"]["traffic"] = hit[0]
                    out["roofline"]["traffic_source"] = os.path.relpath(f, ROOT) + " (FETCH_SIZE x 2, reads only; profile taken at git %s)" % json.load(open(f)).get("git_head", "unrecorded")
                    break
            except Exception:
                pass
    if dtype == "fp8":   # mixed bound, both fractions
        out["roofline_hbm"] = {"bound": "hbm", "achieved": bytes_ / (dom_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": bytes_ / (dom_ms * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_launch": bytes_}
    return out


def kb_figures(seed, k, retrieve_rows=10_548, cold_rows=100_000, d=1536, iters=200):
    """Secondary figures (N = 1; not `value`): the path of reference src/svs/kb.py:1608-1640 end to end through the
    KB mirror (svs_amd.KB: embed lookup -> HIP search -> SQLite fetch of k docs) on an on-disk KB of BASELINE.json
    configs[0]'s size, and the cold start (SQLite file -> first result, reference kb.py:573-618: 98.7 s at 1M rows,
    BASELINE.md) on a `cold_rows`-row KB.  Synthetic unit-norm Gaussian vectors; queries are texts whose vectors
    the embedding function looks up."""
    import shutil
    import tempfile
    import svs_amd
    from svs_amd.kb import _Store

    def write_kb(path, n, rng):
        st = _Store(path)
        with st.transaction():
            for b0 in range(0, n, 20000):
                x = rng.standard_normal((min(20000, n - b0), d)).astype(np.float32)
                x /= np.linalg.norm(x, axis=1, keepdims=True)
                st.conn.executemany("INSERT INTO embeddings (embedding) VALUES (?)", [(r.tobytes(),) for r in x])
                st.conn.executemany("INSERT INTO docs (parent_id, level, text, embedding, meta) VALUES (NULL, 0, ?, ?, NULL)",
                                    [(f"doc {b0 + i}", b0 + i + 1) for i in range(len(x))])
        st.close()

    rng = np.random.default_rng(seed)
    qv = rng.standard_normal((iters + 8, d))
    qv = (qv / np.linalg.norm(qv, axis=1, keepdims=True)).astype(np.float32)
    lookup = {f"query {i}": qv[i].tolist() for i in range(len(qv))}

    async def ef(texts):
        return [lookup[t] for t in texts]

    td = tempfile.mkdtemp(prefix="svs_bench_kb_")
    out = {}
    try:
        p1 = os.path.join(td, "retrieve.sqlite")
        write_kb(p1, retrieve_rows, rng)
        kb = svs_amd.KB(p1, ef)
        kb.load()
        for i in range(iters, iters + 8):
            kb.retrieve(f"query {i}", k)
        lats = []
        for i in range(iters):
            a = time.perf_counter()
            res = kb.retrieve(f"query {i}", k)
            lats.append((time.perf_counter() - a) * 1e3)
        assert len(res) == min(k, retrieve_rows)
        kb.close()
        out["kb_retrieve_p50_ms"] = float(np.median(lats))
        out["kb_retrieve"] = {"rows": retrieve_rows, "dim": d, "k": k, "queries": iters, "min_ms": float(min(lats)),
                              "what": "svs_amd.KB.retrieve() on an on-disk KB: embed lookup + HIP search + SQLite fetch of k docs "
                                      "(the reference publishes 11 ms on a 10,548-doc KB, README.md:128-129)"}
        p2 = os.path.join(td, "cold.sqlite")
        write_kb(p2, cold_rows, rng)
        os.system("sync")
        a = time.perf_counter()
        kb = svs_amd.KB(p2, ef)
        res = kb.retrieve("query 0", k)
        cold = time.perf_counter() - a
        a = time.perf_counter()
        kb.retrieve("query 1", k)
        warm = time.perf_counter() - a
        kb.close()
        out["cold_start_s"] = cold
        out["cold_start"] = {"rows": cold_rows, "dim": d, "file_gb": os.path.getsize(p2) / 1e9, "next_retrieve_ms": warm * 1e3,
                             "what": "open an on-disk KB -> first retrieve() result (BLOBs -> pinned staging blocks -> HBM, "
                                     "svs_index_staging_*); the reference's first query at 1M rows takes 98.7 s (BASELINE.md)"}
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return out


def gen_rows(torch, dev, seed, lo, hi, d, block=62500):
    """Rows [lo, hi) of the synthetic unit-norm Gaussian corpus; block-seeded so
    the content of a row does not depend on how the corpus is sharded."""
    out = torch.empty((hi - lo, d), device=dev, dtype=torch.float32)
    b0, b1 = lo // block, (hi + block - 1) // block
    for b in range(b0, b1):
        g = torch.Generator(device=dev)
        g.manual_seed(seed * 1_000_003 + b)
        x = torch.randn((block, d), device=dev, dtype=torch.float32, generator=g)
        x /= x.norm(dim=1, keepdim=True)
        s, e = max(lo, b * block), min(hi, (b + 1) * block)
        out[s - lo:e - lo] = x[s - b * block:e - b * block]
        del x
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--dtype", choices=["f32", "f16", "fp8"], default="f32",
                    help="HBM element type of the corpus (f32 = the metric's config; f16/fp8 = BASELINE configs[2..4])")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=20)
    ap.add_argument("--latency-iters", type=int, default=200)
    ap.add_argument("--time-every", type=int, default=4,
                    help="record the HIP stage events on every N-th step (1 = every step)")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N > 1: steps whose local top-k records share one RCCL all-gather")
    ap.add_argument("--batch", default="16,256",
                    help="queries per call of the secondary batched figures, comma separated (0: skip)")
    ap.add_argument("--concurrent-solo", action="store_true",
                    help="also time the concurrent callers WITHOUT coalescing")
    ap.add_argument("--concurrent", type=int, default=64,
                    help="threads of the secondary concurrent-callers figure (0: skip)")
    ap.add_argument("--configs", default="2,4",
                    help="BASELINE.json batch configs measured as secondary figures at N = 1 (2: 1M x 1536 f16 x 1024 "
                         "queries; 4: 10M x 3072 fp8 x 256 queries; empty: skip)")
    ap.add_argument("--kb", type=int, default=1,
                    help="1: also time KB.retrieve() end to end on a 10,548-row on-disk KB and the cold start of a "
                         "100,000-row one (N = 1 only; 0: skip)")
    ap.add_argument("--kb-only", action="store_true", help=argparse.SUPPRESS)   # (child mode of --kb: prints kb_figures() as JSON)
    ap.add_argument("--inflight", type=int, default=1,
                    help="searches kept in flight on separate HIP streams (2 lets the top-k "
                         "stage of query i overlap the score stage of query i+1)")
    args = ap.parse_args()

    if args.kb_only:
        print(json.dumps(kb_figures(args.seed + 31, args.k)), flush=True)
        return

    import torch
    import torch.distributed as dist
    from svs_amd import DeviceIndex
    from svs_amd.sharded import ShardedIndex, shard_bounds

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # SVS_BENCH_BACKEND=gloo (rehearsal only): lets several ranks share ONE card on a 1-GPU
    # box to exercise the sharded code path end to end; the real run is RCCL, one GPU per rank.
    backend = os.environ.get("SVS_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # SVS_BENCH_FORCE_DIST=1 (rehearsal only, N = 1): the N > 1 code path -- RCCL process group, records in HBM,
    # one all-gather per `gather_every` steps, async copy home -- with the one rank a 1-GPU box has
    force_dist = world == 1 and bool(os.environ.get("SVS_BENCH_FORCE_DIST"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    elif force_dist:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1, device_id=dev)

    d, k, K, W = args.dim, args.k, args.steps, args.warmup
    if args.scaling == "strong":
        n_total = args.rows
        lo, hi = shard_bounds(n_total, world, rank)
    else:
        n_total = args.rows * world
        lo, hi = rank * args.rows, (rank + 1) * args.rows
    n_local = hi - lo

    # ---- synthetic corpus straight into HBM, then into the index's own layout
    rows = gen_rows(torch, dev, args.seed, lo, hi, d)
    torch.cuda.synchronize()
    idx = DeviceIndex.from_device_pointer(rows.data_ptr(), n_local, d, device=dev_index, row_offset=lo,
                                          dtype=args.dtype)
    if args.variant:
        idx.set_variant(args.variant)
    keep_rows_for_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and args.dtype == "f32"
    if not keep_rows_for_cpu:
        del rows
        torch.cuda.empty_cache()

    # ---- queries: K + W distinct unit vectors, identical on every rank
    g = torch.Generator(device=dev)
    g.manual_seed(args.seed + 77)
    queries = torch.randn((K + W, d), device=dev, dtype=torch.float32, generator=g)
    queries /= queries.norm(dim=1, keepdim=True)

    # The exchange is the LIBRARY's (svs_amd.sharded.ShardedIndex, the same class the gloo tests
    # drive): per step the search kernel writes the packed record [k f32 scores | pad | k i64 rows];
    # N = 1: straight into pinned host memory (zero-copy, no D2H); N > 1: into HBM, one RCCL
    # all-gather per `gather_every` steps on alternating HIP streams, rank 0 streams each gathered
    # chunk home with an async copy and merges (host merge, H1).
    sh = ShardedIndex(idx, n_total, device=dev, gather_every=args.gather_every,
                      streams=(args.inflight if world == 1 else max(2, args.inflight)), force_collective=force_dist)
    G = sh.gather_every
    count = min(k, n_local)
    torch.cuda.synchronize()

    sh.open(K + W, k)          # buffers and streams: outside the timed region
    torch.cuda.synchronize()

    def run_steps(i0, i1):
        """Steps i0 .. i1-1 enqueued (searches + exchanges); collect() drains and merges."""
        for i in range(i0, i1):
            sh.enqueue(queries[i].data_ptr(), d)
        return sh

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warmup (untimed; with the stage events on, so their one-time set-up
    # cost is paid here and not inside the timed region)
    # stage events on every 4th step of the timed region: each timed step carries three event
    # records (~10 us of stream time); sampling keeps the probe from slowing what it measures
    idx.set_timing(0 if os.environ.get("SVS_BENCH_NOEVENTS") else args.time_every)
    run_steps(0, W).collect()
    idx.get_timing()
    barrier()
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps
    t0 = time.perf_counter()
    run_steps(W, W + K)
    t_enq = time.perf_counter()
    results = sh.collect(first=W)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    if os.environ.get("SVS_BENCH_DEBUG"):
        print(f"[rank {rank}] enqueue {1e3*(t_enq-t0):.2f} ms, finish {1e3*(t1-t_enq):.2f} ms", file=sys.stderr)
    score_ms, select_ms, launches = idx.get_timing()
    idx.set_timing(False)

    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- rehearsal check (SVS_BENCH_VERIFY=1, N > 1): the merged result of the sharded run must
    # be IDENTICAL (rows and score bits) to one index over the whole corpus -- the kernels'
    # summation order does not depend on where a row lives
    sharded_check = None
    if world > 1 and rank == 0 and os.environ.get("SVS_BENCH_VERIFY"):
        whole_rows = gen_rows(torch, dev, args.seed, 0, n_total if args.scaling == "strong" else args.rows * world, d)
        whole = DeviceIndex.from_device_pointer(whole_rows.data_ptr(), whole_rows.shape[0], d, device=dev_index, dtype=args.dtype)
        del whole_rows
        bad = 0
        for j in range(min(16, K)):
            exp = whole.search(queries[W + j].cpu().numpy(), k)
            got_s, got_r = results[j]
            if [int(x) for x in got_r] != [r for _, r in exp] or [float(x) for x in got_s] != [sc for sc, _ in exp]:
                bad += 1
        whole.release()
        sharded_check = {"queries": min(16, K), "mismatches": bad}

    # ---- p50 latency at the C-ABI boundary (host buffers in, results out, synced): its own
    # latency_iters (>= 200) distinct queries, whatever --steps / --warmup were
    lat_ms, lat_n = None, 0
    if world == 1:
        gl = torch.Generator(device=dev)
        gl.manual_seed(args.seed + 555)
        ql = torch.randn((max(args.latency_iters, 200) + 5, d), device=dev, dtype=torch.float32, generator=gl)
        qh = (ql / ql.norm(dim=1, keepdim=True)).cpu().numpy()
        del ql
        for q in qh[:5]:
            idx.search(q, k)
        lats = []
        for q in qh[5:]:
            a = time.perf_counter()
            idx.search(q, k)
            lats.append((time.perf_counter() - a) * 1e3)
        lat_ms, lat_n = float(np.median(lats)), len(lats)

    # HBM traffic of the dominant kernel from the committed PMC passes (separate
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950
    # corrections applied by tools/summarize_prof.py); only quoted for the
    # configuration it was collected on.
    traffic, traffic_src = None, None
    if rank == 0 and n_local == 1_000_000 and d == 1536:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))[::-1]:
            try:
                pm = json.load(open(f))["pmc"]
                kn = [v for kname, v in pm.items() if "gemv_f32" in kname and "hbm_bytes_per_launch" in v]
                if kn:
                    traffic = kn[0]["hbm_bytes_per_launch"]
                    traffic_src = os.path.relpath(f, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; profile taken at git %s)" % json.load(open(f)).get("git_head", "unrecorded")
                    break
            except Exception:
                continue

    # ---- secondary figure (N = 1 only; not `value`): batched search, 16 queries
    # share one pass over the corpus (north_star: >= 10,000 queries/s at 1 GPU)
    batched = None
    if world == 1 and args.batch:
        batched = []
        for B in [int(x) for x in str(args.batch).split(",") if int(x) > 1]:
            gb = torch.Generator(device=dev)
            gb.manual_seed(args.seed + 999 + B)
            qb = torch.randn((B, d), device=dev, dtype=torch.float32, generator=gb)
            qb = (qb / qb.norm(dim=1, keepdim=True)).cpu().numpy()
            # (the first ~10 ms after a change of kernel mix run at a settling clock: the 16-query
            # kernel takes 1.0-1.3 ms during it and 1.0 after, profiles/r1_b16_kernel_stats.csv)
            for _ in range(max(3, min(15, 256 // B))):
                idx.search_batch(qb, k)
            idx.set_timing(True)
            reps = max(3, min(30, 512 // B))
            a = time.perf_counter()
            for _ in range(reps):
                idx.search_batch(qb, k)
            dt = time.perf_counter() - a
            b_score, b_sel, b_cnt = idx.get_timing()
            idx.set_timing(False)
            batched.append({"queries_per_call": B, "value": B * reps / dt, "unit": "queries/s",
                            "ms_per_call": dt / reps * 1e3, "score_ms": b_score / max(b_cnt, 1),
                            "select_ms": b_sel / max(b_cnt, 1),
                            "note": "host API (queries in, results out, synchronised)"})
    idx.set_timing(False)

    # ---- secondary figure (N = 1 only; not `value`): CONCURRENT single-query callers, the way AsyncKB.retrieve
    # meets the search (one executor thread per task, reference src/svs/kb.py:1184-1190): without and with
    # svs_index_set_coalesce.  Python threads, so the GIL caps it (~12 k queries/s; 64 C threads reach 24.6 k:
    # tools/coalesce_bench.c)
    concurrent = None
    if world == 1 and args.concurrent > 1 and n_local * d:
        import threading
        gc = torch.Generator(device=dev)
        gc.manual_seed(args.seed + 4242)
        qc = torch.randn((256, d), device=dev, dtype=torch.float32, generator=gc)
        qc = (qc / qc.norm(dim=1, keepdim=True)).cpu().numpy()
        concurrent = {"callers": args.concurrent, "unit": "queries/s", "note": "Python threads, one query per call, host API"}
        # (the solo figure is opt-in: 64 uncoordinated callers put 8 score kernels on the card at once, each
        #  several times slower than alone -- that would sit in the kernel statistics of the dominant kernel that
        #  profiles/ records for this very command; alone they get `value`, ~1,140 queries/s, whatever their number)
        for mode in (("solo", "coalesced") if args.concurrent_solo else ("coalesced",)):
            idx.set_coalesce(mode == "coalesced")
            p0, a0 = idx.coalesce_stats()
            done = [0] * args.concurrent
            stop = time.time() + 1.5

            def caller(t):
                i = t
                while time.time() < stop:
                    idx.search(qc[i % 256], k)
                    i += args.concurrent
                    done[t] += 1
            ts = [threading.Thread(target=caller, args=(t,)) for t in range(args.concurrent)]
            t0 = time.time()
            [t.start() for t in ts]
            [t.join() for t in ts]
            dt = time.time() - t0
            concurrent[mode] = sum(done) / dt
            concurrent[mode + "_mean_latency_ms"] = 1e3 * dt * args.concurrent / max(sum(done), 1)
            if mode == "coalesced":
                p1, a1 = idx.coalesce_stats()
                concurrent["queries_per_corpus_pass"] = (a1 - a0) / max(p1 - p0, 1)
        idx.set_coalesce(False)
        # the same protocol from C threads (no GIL): tools/coalesce_bench.c, built by __graft_entry__.build(), run as
        # a child process on its own 1M x 1536 index (a pure-C caller of the ABI: svs_index_create + svs_index_search)
        exe = os.path.join(ROOT, "tools", "coalesce_bench_c")
        if os.path.exists(exe) and args.dtype == "f32":
            import subprocess
            try:
                r = subprocess.run([exe, str(n_local), str(d), "1.5", str(args.concurrent), "1"], capture_output=True, text=True, timeout=300)
                for line in r.stdout.splitlines():
                    if line.startswith("RESULT "):
                        _, T, mode, qps, lat, qpp = line.split()
                        concurrent["c_threads"] = {"callers": int(T), "coalesced": float(qps), "coalesced_mean_latency_ms": float(lat),
                                                   "queries_per_corpus_pass": float(qpp), "note": "pthreads calling svs_index_search(nq = 1), tools/coalesce_bench.c"}
                if "c_threads" not in concurrent:
                    concurrent["c_threads"] = {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as e:   # noqa: BLE001 -- a secondary figure must not sink the bench line
                concurrent["c_threads"] = {"error": repr(e)[:300]}

    # ---- secondary figures (N = 1 only; not `value`): BASELINE.json configs[2] and configs[4], the
    # MFMA-bound batch configurations, each on its own index (synthetic, same recipe as the headline corpus)
    configs = []
    want = [c for c in str(args.configs).split(",") if c.strip()] if (world == 1 and args.dtype == "f32") else []
    if "2" in want and n_total == 1_000_000 and d == 1536:
        src = rows if keep_rows_for_cpu else gen_rows(torch, dev, args.seed, 0, n_total, d)
        i2 = DeviceIndex.from_device_pointer(src.data_ptr(), n_total, d, device=dev_index, dtype="f16")
        torch.cuda.synchronize()
        if not keep_rows_for_cpu:
            del src
        configs.append(batched_config(torch, i2, "BASELINE.json configs[2]: 1M docs x dim 1536 fp16, top-100, batch 1024 queries, "
                                      "MFMA GEMM + fused top-k", n_total, d, "f16", 1024, k, args.seed + 2, reps=10))
        i2.release()
        torch.cuda.empty_cache()
    if "4" in want:
        n4, d4, blk = 10_000_000, 3072, 500_000
        i4 = DeviceIndex.empty(d4, device=dev_index, dtype="fp8", reserve=n4)
        for b0 in range(0, n4, blk):           # the 123 GB f32 source never exists: one 6 GB block at a time
            x = gen_rows(torch, dev, args.seed + 4, b0, b0 + blk, d4, block=blk)
            torch.cuda.synchronize()
            i4.append_device(x.data_ptr(), blk)
            del x
        torch.cuda.empty_cache()
        configs.append(batched_config(torch, i4, "BASELINE.json configs[4]: 10M docs x dim 3072 fp8 (e4m3 + row scales), top-100, "
                                      "batch 256 queries, fp8 MFMA GEMM + fused top-k", n4, d4, "fp8", 256, k, args.seed + 4, reps=5))
        i4.release()
        torch.cuda.empty_cache()

    # (in a child process: its small-corpus launches of the single-query kernels would otherwise sit in the
    #  per-kernel statistics rocprofv3 keeps for THIS process, next to the 1M-row launches the roofline is about)
    kb_out = {}
    if world == 1 and args.kb and args.dtype == "f32":
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--kb-only", "--seed", str(args.seed), "--k", str(k)],
                               capture_output=True, text=True, timeout=600)
            kb_out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:   # noqa: BLE001 -- a secondary figure must not sink the bench line
            kb_out = {"kb_retrieve": {"error": repr(e)[:300]}}

    out = None
    if rank == 0:
        kernel_ms = score_ms / max(launches, 1)
        esz = {"f32": 4, "f16": 2, "fp8": 1}[args.dtype]
        alg_bytes = float(n_local) * d * esz + (4.0 * n_local if args.dtype == "fp8" else 0.0)
        achieved = alg_bytes / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
        out = {
            "metric": "queries/sec, cosine top-%d over %dx%d %s, single query" % (k, n_total, d, {"f32": "fp32", "f16": "fp16", "fp8": "fp8"}[args.dtype]),
            "value": K / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE.json configs[1]: " if (args.dtype == "f32" and n_total == 1_000_000 and d == 1536) else "") +
                            "%d docs x dim %d " % (n_total, d) + args.dtype + ", top-%d, single query, " % k +
                            "HBM-resident GEMV + top-k" + ("" if world == 1 else ", row-sharded over %d GPUs + RCCL all-gather + host merge" % world),
                "rows_per_gpu": n_local, "dim": d, "k": k, "queries_per_step": 1,
                "corpus": "unit-norm gaussian, seed %d, generated on device" % args.seed,
                "variant": args.variant, "searches_in_flight": sh.streams,
                "steps_per_exchange": G if (world > 1 or force_dist) else None,
            },
            "p50_latency_ms": lat_ms,
            "p50_latency_queries": lat_n,
            "kb_retrieve_p50_ms": kb_out.get("kb_retrieve_p50_ms"),
            "cold_start_s": kb_out.get("cold_start_s"),
            "kb": {k_: v for k_, v in kb_out.items() if k_ in ("kb_retrieve", "cold_start")} or None,
            "sharded_check": sharded_check,
            "batched": batched,
            "concurrent": concurrent,
            "configs": configs,
            "stage_ms": {"score": kernel_ms, "select": select_ms / max(launches, 1)},
            "roofline": {
                "bound": "hbm", "kernel": KERNEL_NAME[(args.dtype, 1)] if (n_local * d) else "-",
                "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes, "launches_timed": launches,
                "timed_every_nth_step": args.time_every,
                "avg_launch_ms": kernel_ms,
            },
        }

    # ---- CPU baseline: the numpy restatement of the reference path, this host's cores
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.dtype == "f32":
        from oracle import svs_oracle as oracle  # cpu_baseline leg only (checker + baseline)
        m_host = rows.cpu().numpy()
        del rows
        torch.cuda.empty_cache()
        # its own 3 + cpu_iters queries (the driver's --steps 20 --warmup 5 leaves only 25 in `queries`)
        gc = torch.Generator(device=dev)
        gc.manual_seed(args.seed + 4242)
        qh = torch.randn((3 + args.cpu_iters, d), device=dev, dtype=torch.float32, generator=gc)
        qh = (qh / qh.norm(dim=1, keepdim=True)).cpu().numpy()
        for q in qh[:3]:
            oracle.cpu_search(m_host, q, k)
        ts = []
        for q in qh[3:3 + args.cpu_iters]:
            a = time.perf_counter()
            oracle.cpu_search(m_host, q, k)
            ts.append(time.perf_counter() - a)
        p50 = float(np.median(ts))
        # parity spot check of the timed results against the oracle (not timed)
        mism = 0
        for j in range(min(8, K)):
            exp = oracle.cpu_search(m_host, queries[W + j].cpu().numpy(), k)
            got_s, got_r = results[j]
            if [int(x) for x in got_r] != [i for _, i in exp] or \
                    max(abs(float(a) - b) for a, (b, _) in zip(got_s, exp)) > 1e-5:
                mism += 1
        blas = "threadpoolctl unavailable"
        try:
            import threadpoolctl
            info = threadpoolctl.threadpool_info()
            thr = max([p.get("num_threads", 1) for p in info] or [1])
            blas = "; ".join("%s %s (%s, %s threads)" % (p.get("internal_api"), p.get("version"), p.get("threading_layer", "-"), p.get("num_threads"))
                             for p in info if p.get("user_api") == "blas") or "no BLAS pool reported"
        except Exception:
            thr = os.cpu_count()
        blas += "; OPENBLAS_NUM_THREADS=%s OMP_NUM_THREADS=%s, os.cpu_count()=%s" % (
            os.environ.get("OPENBLAS_NUM_THREADS", "unset"), os.environ.get("OMP_NUM_THREADS", "unset"), os.cpu_count())
        try:
            thr = min(int(thr), len(os.sched_getaffinity(0)))
        except Exception:
            pass
        out["cpu_baseline"] = {
            "value": 1.0 / p50, "unit": "queries/s", "cores": int(thr), "kind": "port",
            "sample": "full %dx%d corpus, %d queries after 3 warm-ups, numpy %s np.dot + argpartition + sort (oracle/svs_oracle.py); p50 %.2f ms, min %.2f ms; BLAS: %s"
                      % (n_total, d, len(ts), np.__version__, p50 * 1e3, min(ts) * 1e3, blas),
        }
        out["parity_spot_check"] = {"queries": min(8, K), "mismatches": mism}

    if rank == 0:
        print(json.dumps(out), flush=True)
    idx.release()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
