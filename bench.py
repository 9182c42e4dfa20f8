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
  python bench.py --gpus N ...            (no launcher: starts its own N rank processes, see self_launch)
  python bench.py --gpus N --config 3     (BASELINE.json configs[3]: 12.5M x 1536 f16 rows PER GPU, weak scaling)
  python bench.py --gpus N --multi        (the same shards through svs_multi_search: one process, no RCCL)

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


def child_env(extra=None):
    """Environment for child processes: without the profiler's preload.  Under `rocprofv3 --pmc -- python3 bench.py`
    every child would inherit a preloaded library that initialises the GPU before the child's own program runs,
    and a child that then execs another program (sh -c ...) is the exec hop this pool forbids."""
    env = {k: v for k, v in os.environ.items()
           if not (k == "LD_PRELOAD" or k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTRACER", "ROCTX")))}
    env.update(extra or {})
    return env


def profile_traffic(pattern, kernel_substr, field):
    """HBM bytes per launch of a kernel from the newest committed PMC summary under profiles/ whose `csrc_sha16`
    stamp equals the kernel sources this run was built from (svs_amd/buildinfo.py); a summary of other sources is
    refused: (None, why)."""
    from svs_amd.buildinfo import csrc_sha16
    here = csrc_sha16()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            js = json.load(open(f))
            hit = [v[field] for kn, v in js.get("pmc", {}).items() if kernel_substr in kn and field in v]
            if not hit:
                continue
            if js.get("csrc_sha16") != here:
                stale = stale or "%s is of other kernel sources (csrc_sha16 %s, this tree %s): not quoted" % (
                    os.path.relpath(f, ROOT), js.get("csrc_sha16", "unstamped"), here)
                continue
            return hit[0], "%s (rocprofv3 --pmc passes of tools/profile_round.sh, gfx950 corrections of tools/summarize_*.py; csrc_sha16 %s = this tree)" % (
                os.path.relpath(f, ROOT), here)
        except Exception:
            continue
    return None, stale or "no PMC summary under profiles/ for this kernel"


def self_launch(args):
    """`python bench.py --gpus N` with no launcher: N fresh rank processes of this same file (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), rank 0's JSON line relayed.  This process has not imported
    torch or touched the GPU; the ranks are spawned children, never an exec of an initialised process."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = child_env({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "MASTER_ADDR": "127.0.0.1",
                         "MASTER_PORT": str(port), "SVS_BENCH_SELF_LAUNCHED": "1"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    deadline = time.time() + float(os.environ.get("SVS_BENCH_LAUNCH_TIMEOUT", "1500"))
    failed = None
    while any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0) and failed is None:
                failed = (r, p.returncode)
                deadline = min(deadline, time.time() + 20)   # the others may be stuck in a collective: a short grace, then stop them
        if time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()                                  # (exact children of this process)
            break
        if procs[0].poll() is not None and failed is None and all(p.poll() is not None for p in procs):
            break
        time.sleep(0.05)
    out0 = procs[0].stdout.read().decode("utf-8", "replace") if procs[0].stdout else ""
    for p in procs:
        p.wait()
    sys.stdout.write(out0)
    sys.stdout.flush()
    rcs = [p.returncode for p in procs]
    if failed or any(rcs):
        print("bench.py self-launch: rank exit codes %s" % rcs, file=sys.stderr)
        sys.exit(1)
    sys.exit(0)


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
    per_call = []
    for _ in range(reps):
        a = time.perf_counter()
        idx.search_batch(q, k)
        per_call.append(time.perf_counter() - a)
    # (the MEDIAN call: one host hiccup among `reps` synchronous calls -- a 10 ms stall was seen once on a shared box -- is
    #  not the library's rate; mean and slowest call are reported beside it)
    dt = float(np.median(per_call)) * reps
    score_ms, select_ms, cnt = idx.get_timing()
    dom_ms = idx.last_dominant_ms_sum / max(cnt, 1)
    idx.set_timing(False)
    # two callers in flight on the one handle (each its own thread and batch; the entry points are re-entrant, ctypes
    # releases the GIL): call N + 1's host copy, upload and prefix pass run under call N's whole-corpus GEMM
    import threading
    q2 = np.ascontiguousarray(q[::-1])
    done = [0, 0]

    def caller(t, n_calls, count):
        for _ in range(n_calls):
            idx.search_batch(q if t == 0 else q2, k)
            if count:
                done[t] += 1
    # (one untimed round first: the second caller's search context -- stream, pinned buffers, 0.3 GB of candidate scratch -- is made on its first call)
    th = [threading.Thread(target=caller, args=(t, 2, False)) for t in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    th = [threading.Thread(target=caller, args=(t, reps, True)) for t in range(2)]
    a = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt2 = time.perf_counter() - a
    esz = {"f32": 4, "f16": 2, "fp8": 1}[dtype]
    flops = 2.0 * n * d * nq
    bytes_ = float(n) * d * esz + (4.0 * n if dtype == "fp8" else 0.0)
    pf = flops / (dom_ms * 1e-3)
    out = {"workload": name, "queries_per_call": nq, "ms_per_call": dt / reps * 1e3, "value": nq * reps / dt, "unit": "queries/s",
           "calls_timed": reps, "ms_per_call_mean": float(np.mean(per_call)) * 1e3, "ms_per_call_max": float(np.max(per_call)) * 1e3,
           "stage_ms": {"score": score_ms / max(cnt, 1), "select": select_ms / max(cnt, 1), "dominant_kernel": dom_ms},
           "roofline": {"bound": "mfma", "kernel": KERNEL_NAME.get((dtype, nq), "batched GEMM"), "achieved": pf / 1e12,
                        "peak": MFMA_PEAK[dtype] / 1e12, "unit": "TFLOP/s", "frac": pf / MFMA_PEAK[dtype],
                        "algorithmic_flops_per_launch": flops, "avg_launch_ms": dom_ms, "launches_timed": cnt,
                        "traffic": None},
           "two_callers_in_flight": {"value": nq * sum(done) / dt2, "unit": "queries/s", "ms_per_call_and_caller": dt2 / reps * 1e3,
                                     "note": "two threads, each calling svs_index_search with its own batch on the same handle"},
           "note": "host API (queries in, results out, synchronised); ms_per_call / value = the median of calls_timed calls; kernel time from HIP events inside the library"}
    # HBM bytes of that kernel from the committed FETCH_SIZE pass (tools/profile_round.sh) -- only from a profile of
    # THESE kernel sources
    cfg_tag = {("f16", 1024): "cfg2", ("fp8", 256): "cfg4"}.get((dtype, nq))
    if cfg_tag:
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = profile_traffic(
            f"r*_{cfg_tag}_summary.json", "gemm_phased_kernel<true", "hbm_read_bytes_per_launch")
    if dtype == "fp8":   # mixed bound, both fractions
        out["roofline_hbm"] = {"bound": "hbm", "achieved": bytes_ / (dom_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                               "frac": bytes_ / (dom_ms * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_launch": bytes_}
    return out


def kb_figures(seed, k, retrieve_rows=10_548, cold_rows=100_000, big_rows=1_000_000, d=1536, iters=200, callers=64):
    """Secondary figures (N = 1; not `value`): the path of reference src/svs/kb.py:1608-1640 end to end through the
    KB mirror (svs_amd.KB: embed lookup -> HIP search -> SQLite fetch of k docs) on an on-disk KB of BASELINE.json
    configs[0]'s size; the reference's ASYNC surface under load (src/svs/kb.py:1171-1206: `callers` concurrent
    AsyncKB.retrieve() tasks -- executor hop, coalesced search, SQLite fetch under the lock) beside the same
    queries through one retrieve_many; the cold start (SQLite file -> first result, reference kb.py:573-618: 98.7 s
    at 1M rows) on a `cold_rows`-row KB; and, when `big_rows` > 0 and the disk takes it inside the
    budget, the reference's own headline -- KB.retrieve() on a 1M-row on-disk KB (README.md:129: 0.24 s) -- with
    its cold start.  Synthetic unit-norm Gaussian vectors; queries are texts whose vectors the embedding function
    looks up."""
    import asyncio
    import shutil
    import tempfile
    import svs_amd
    from svs_amd.kb import _Store

    def write_kb(path, n, rng, block=20000):
        st = _Store(path)
        with st.transaction():
            for b0 in range(0, n, block):
                x = rng.standard_normal((min(block, n - b0), d), dtype=np.float32)
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

    def retrieve_p50(kb, n_q):
        for i in range(iters, iters + 8):
            kb.retrieve(f"query {i}", k)
        lats = []
        for i in range(n_q):
            a = time.perf_counter()
            res = kb.retrieve(f"query {i}", k)
            lats.append((time.perf_counter() - a) * 1e3)
        return lats, res

    td = tempfile.mkdtemp(prefix="svs_bench_kb_")
    out = {}
    try:
        p1 = os.path.join(td, "retrieve.sqlite")
        write_kb(p1, retrieve_rows, rng)
        kb = svs_amd.KB(p1, ef)
        kb.load()
        lats, res = retrieve_p50(kb, iters)
        assert len(res) == min(k, retrieve_rows)
        kb.close()
        out["kb_retrieve_p50_ms"] = float(np.median(lats))
        out["kb_retrieve"] = {"rows": retrieve_rows, "dim": d, "k": k, "queries": iters, "min_ms": float(min(lats)),
                              "what": "svs_amd.KB.retrieve() on an on-disk KB: embed lookup + HIP search + SQLite fetch of k docs "
                                      "(the reference publishes 11 ms on a 10,548-doc KB, README.md:128-129)"}

        # ---- the async surface under load: `callers` AsyncKB.retrieve() tasks in flight at once, in waves
        async def drive():
            akb = svs_amd.AsyncKB(p1, ef)
            await akb.load()
            names = [f"query {i}" for i in range(callers)]
            await asyncio.gather(*[akb.retrieve(q, k) for q in names[:8]])
            idx = getattr(akb.embeddings_matrix, "index", None)
            st0 = idx.coalesce_stats() if idx is not None and hasattr(idx, "coalesce_stats") else None
            lat = []

            async def one(q):
                a = time.perf_counter()
                r = await akb.retrieve(q, k)
                lat.append((time.perf_counter() - a) * 1e3)
                return r
            waves = 6
            a = time.perf_counter()
            for _ in range(waves):
                got = await asyncio.gather(*[one(q) for q in names])
            dt = time.perf_counter() - a
            st1 = idx.coalesce_stats() if st0 is not None else None
            await akb.retrieve_many(names, k)
            b = time.perf_counter()
            reps = 6
            for _ in range(reps):
                many = await akb.retrieve_many(names, k)
            dm = (time.perf_counter() - b) / reps
            same = all([x["doc"]["id"] for x in g1] == [x["doc"]["id"] for x in g2] for g1, g2 in zip(got, many))
            await akb.close()
            r = {"callers": callers, "rows": retrieve_rows, "k": k, "queries_per_s": waves * callers / dt,
                 "p50_ms": float(np.median(lat)), "max_ms": float(max(lat)),
                 "retrieve_many": {"queries": callers, "ms_per_call": dm * 1e3, "queries_per_s": callers / dm,
                                   "same_docs_as_the_concurrent_retrieves": bool(same)},
                 "what": "asyncio.gather of %d AsyncKB.retrieve() tasks per wave, %d waves (reference src/svs/kb.py:1171-1206: embed, "
                         "search on an executor thread outside the lock -- coalesced in the library --, k docs fetched from SQLite "
                         "under the lock), beside ONE AsyncKB.retrieve_many of the same queries (one batched search, one IN (...) "
                         "fetch per query)" % (callers, waves)}
            if st0 is not None:
                r["queries_per_corpus_pass"] = (st1[1] - st0[1]) / max(st1[0] - st0[0], 1)
            return r
        try:
            out["async_retrieve"] = asyncio.run(drive())
        except Exception as e:   # noqa: BLE001 -- a secondary figure must not sink the others
            out["async_retrieve"] = {"error": repr(e)[:300]}

        def cold(path, rows):
            os.sync()
            a = time.perf_counter()
            kb = svs_amd.KB(path, ef)
            res = kb.retrieve("query 0", k)
            cold_s = time.perf_counter() - a
            assert len(res) == min(k, rows)
            return kb, cold_s

        p2 = os.path.join(td, "cold.sqlite")
        t_w = time.perf_counter()
        write_kb(p2, cold_rows, rng)
        t_w = time.perf_counter() - t_w
        kb, cold_s = cold(p2, cold_rows)
        a = time.perf_counter()
        kb.retrieve("query 1", k)
        warm = time.perf_counter() - a
        kb.close()
        os.remove(p2)
        out["cold_start_s"] = cold_s
        out["cold_start"] = {"rows": cold_rows, "dim": d, "next_retrieve_ms": warm * 1e3,
                             "what": "open an on-disk KB -> first retrieve() result (BLOBs -> pinned staging blocks -> HBM, "
                                     "svs_index_staging_*); the reference's first query at 1M rows takes 98.7 s (BASELINE.md)"}
        # ---- the reference's headline: a 1M-row KB on disk (6.2 GB of BLOBs).  Written only if the 100k-row write
        # says it fits ~75 s and the temp dir has the room.
        free = shutil.disk_usage(td).free
        if big_rows and t_w * (big_rows / cold_rows) < 75.0 and free > 9e9 * (big_rows / 1e6):
            p3 = os.path.join(td, "big.sqlite")
            a = time.perf_counter()
            write_kb(p3, big_rows, rng, block=50000)
            t_write = time.perf_counter() - a
            kb, cold_big = cold(p3, big_rows)
            lats, res = retrieve_p50(kb, 100)
            kb.close()
            out["kb_retrieve_1m_p50_ms"] = float(np.median(lats))
            out["kb_1m"] = {"rows": big_rows, "dim": d, "k": k, "queries": len(lats), "min_ms": float(min(lats)),
                            "cold_start_s": cold_big, "file_gb": os.path.getsize(p3) / 1e9, "write_s": t_write,
                            "what": "svs_amd.KB on a 1M-row on-disk KB: cold_start_s = open -> first retrieve() result (reference: 98.7 s, "
                                    "examples/One Million Documents Benchmark.ipynb:236-237), p50 of the next 100 retrieve() calls end to "
                                    "end (reference: 0.24 s, README.md:129)"}
        else:
            out["kb_1m"] = {"skipped": "writing %d rows would take ~%.0f s here (disk free %.0f GB): see tools/kb_coldstart.py"
                                       % (big_rows, t_w * (big_rows / cold_rows), free / 1e9)}
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


def build_index(torch, dev, dev_index, seed, lo, hi, d, dtype, blk=500_000):
    """Rows [lo, hi) of the synthetic corpus as a DeviceIndex with row_offset = lo, built block by block on the
    device (gen_rows -> svs_index_append_from_device): the f32 source of a 12.5M x 1536 shard (76.8 GB) or of the
    10M x 3072 fp8 corpus (123 GB) never exists at once."""
    from svs_amd import DeviceIndex
    idx = DeviceIndex.empty(d, device=dev_index, row_offset=lo, dtype=dtype, reserve=hi - lo)
    for b0 in range(lo, hi, blk):
        x = gen_rows(torch, dev, seed, b0, min(hi, b0 + blk), d)
        torch.cuda.synchronize()
        idx.append_device(x.data_ptr(), x.shape[0])
        del x
    torch.cuda.empty_cache()
    return idx


def single_query_config(torch, idx, name, n, d, dtype, k, seed, iters=60):
    """Secondary figure: one GPU's share of a single-query configuration (BASELINE.json configs[3]) on an index already
    in HBM: host-API latency of `iters` distinct queries, score kernel time from HIP events inside the library."""
    dev = torch.device("cuda", idx.device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    q = torch.randn((iters + 5, d), device=dev, dtype=torch.float32, generator=g)
    q = (q / q.norm(dim=1, keepdim=True)).cpu().numpy()
    for x in q[:5]:
        idx.search(x, k)
    idx.set_timing(True)
    lats = []
    for x in q[5:]:
        a = time.perf_counter()
        idx.search(x, k)
        lats.append((time.perf_counter() - a) * 1e3)
    score_ms, select_ms, cnt = idx.get_timing()
    idx.set_timing(False)
    esz = {"f32": 4, "f16": 2, "fp8": 1}[dtype]
    kms = score_ms / max(cnt, 1)
    bytes_ = float(n) * d * esz + (4.0 * n if dtype == "fp8" else 0.0)
    return {"workload": name, "queries_per_call": 1, "ms_per_call": float(np.median(lats)), "value": 1e3 / float(np.median(lats)),
            "unit": "queries/s", "stage_ms": {"score": kms, "select": select_ms / max(cnt, 1), "dominant_kernel": kms},
            "roofline": {"bound": "hbm", "kernel": KERNEL_NAME.get((dtype, 1), "single-query GEMV"), "achieved": bytes_ / (kms * 1e-3) / 1e9,
                         "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bytes_ / (kms * 1e-3) / HBM_PEAK,
                         "algorithmic_bytes_per_launch": bytes_, "avg_launch_ms": kms, "launches_timed": cnt, "traffic": None},
            "note": "host API, one query per call (p50 of %d); at N GPUs every GPU holds such a shard (weak scaling: "
                    "python bench.py --gpus N --config 3)" % len(lats)}


def run_multi(args):
    """`--multi`: the shards of the N-rank run inside ONE process through the multi-device entry -- config 1: the C ABI's
    svs_multi_create / svs_multi_search (NativeMultiIndex: host matrix in, one worker thread per shard inside the library,
    merge on the caller's thread); config 3: MultiDeviceIndex over shards built on their devices block by block (a 100M-row
    f32 host matrix does not exist).  Shard g lives on device g % visible devices.  Same JSON line, `config.path` says which."""
    import torch
    from svs_amd import DeviceIndex
    from svs_amd.multi import MultiDeviceIndex, NativeMultiIndex
    from svs_amd.sharded import shard_bounds
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    G, d, k, K, W = args.gpus, args.dim, args.k, args.steps, args.warmup
    ndev = torch.cuda.device_count()
    devices = [g % ndev for g in range(G)]
    dev0 = torch.device("cuda", 0)
    if args.scaling == "strong":
        n_total = args.rows
        bounds = [shard_bounds(n_total, G, g) for g in range(G)]
    else:
        n_total = args.rows * G
        bounds = [(g * args.rows, (g + 1) * args.rows) for g in range(G)]
    if args.config == 1:
        host = gen_rows(torch, dev0, args.seed, 0, n_total, d).cpu().numpy()
        idx = NativeMultiIndex(host, devices=devices, dtype=args.dtype)
        del host
        path = "svs_multi_create / svs_multi_search (C ABI, one process, %d shards)" % G
    else:
        shards = [build_index(torch, torch.device("cuda", devices[g]), devices[g], args.seed, lo, hi, d, args.dtype)
                  for g, (lo, hi) in enumerate(bounds)]
        idx = MultiDeviceIndex(None, _shards=shards, _bounds=bounds)
        path = "svs_amd.MultiDeviceIndex (one process, one thread per shard, %d shards built on their devices)" % G
    g = torch.Generator(device=dev0)
    g.manual_seed(args.seed + 77)
    queries = torch.randn((K + W, d), device=dev0, dtype=torch.float32, generator=g)
    queries = (queries / queries.norm(dim=1, keepdim=True)).cpu().numpy()
    for i in range(W):
        idx.search(queries[i], k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(W, W + K):
        res = idx.search(queries[i], k)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert len(res) == min(k, n_total)
    esz = {"f32": 4, "f16": 2, "fp8": 1}[args.dtype]
    out = {"metric": "queries/sec, cosine top-%d over %dx%d %s, single query" % (k, n_total, d, {"f32": "fp32", "f16": "fp16", "fp8": "fp8"}[args.dtype]),
           "value": K / elapsed, "unit": "queries/s", "n_gpus": G, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
           "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": (("BASELINE.json configs[%d]: " % (3 if args.config == 3 else 1))
                                   if ((args.config == 3 and args.rows == 12_500_000 and d == 1536) or (args.config == 1 and n_total == 1_000_000 and d == 1536 and args.dtype == "f32")) else "")
                                  + "%d docs x dim %d %s, top-%d, single query, "
                                  "row-sharded over %d shards in ONE process + host merge" % (n_total, d, args.dtype, k, G),
                      "path": path, "devices": devices, "rows_per_shard": [hi - lo for lo, hi in bounds], "dim": d, "k": k},
           "roofline": {"bound": "hbm", "kernel": KERNEL_NAME[(args.dtype, 1)], "unit": "GB/s", "peak": HBM_PEAK / 1e9 * len(set(devices)),
                        "achieved": float(n_total) * d * esz / (elapsed / K) / 1e9,
                        "frac": float(n_total) * d * esz / (elapsed / K) / (HBM_PEAK * len(set(devices))), "traffic": None,
                        "note": "whole step (all shards in parallel + merge) against the visible devices' combined HBM peak: an end-to-end "
                                "fraction, not a kernel's; the per-kernel figure is the one-process-per-GPU run's"},
           "cpu_baseline": None}
    print(json.dumps(out), flush=True)
    idx.release()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None,
                    help="default: strong for --config 1 (the 1M-row corpus is split), weak for --config 3")
    ap.add_argument("--config", type=int, choices=[1, 3], default=1,
                    help="BASELINE.json config measured as `value`: 1 = 1M x 1536 f32, single query (the metric's config); "
                         "3 = 12.5M x 1536 f16 rows PER GPU, single query (100M rows on 8 GPUs; weak scaling)")
    ap.add_argument("--multi", action="store_true",
                    help="one process, --gpus shards through the multi-device entry (svs_multi_search / MultiDeviceIndex): the "
                         "same shards as the one-process-per-GPU run, no torch.distributed, no RCCL (cross-check of the curve)")
    ap.add_argument("--dtype", choices=["f32", "f16", "fp8"], default=None,
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
    ap.add_argument("--configs", default="2,3,4",
                    help="BASELINE.json configs measured as secondary figures at N = 1 (2: 1M x 1536 f16 x 1024 "
                         "queries; 3: one GPU's share of configs[3], 12.5M x 1536 f16, single query; 4: 10M x 3072 fp8 x 256 "
                         "queries; empty: skip)")
    ap.add_argument("--kb", type=int, default=1,
                    help="1: also time KB.retrieve() end to end on a 10,548-row on-disk KB and the cold start of a "
                         "100,000-row one (N = 1 only; 0: skip)")
    ap.add_argument("--kb-only", action="store_true", help=argparse.SUPPRESS)   # (child mode of --kb: prints kb_figures() as JSON)
    ap.add_argument("--inflight", type=int, default=1,
                    help="searches kept in flight on separate HIP streams (2 lets the top-k "
                         "stage of query i overlap the score stage of query i+1)")
    args = ap.parse_args()
    if args.config == 3:
        args.rows = 12_500_000 if args.rows == 1_000_000 else args.rows
        args.dtype = args.dtype or "f16"
        args.scaling = args.scaling or "weak"
    args.dtype = args.dtype or "f32"
    args.scaling = args.scaling or "strong"

    if args.kb_only:
        print(json.dumps(kb_figures(args.seed + 31, args.k)), flush=True)
        return
    if args.multi:
        return run_multi(args)
    # no launcher (the shape of the N = 1 command): start the ranks ourselves, before anything touches the GPU
    if args.gpus > 1 and "RANK" not in os.environ:
        return self_launch(args)

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
    main_cfg = args.config == 1 and args.dtype == "f32"           # the metric's config: carries the secondary figures
    keep_rows_for_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and main_cfg
    rows = None
    if keep_rows_for_cpu or (hi - lo) * d * 4 <= (8 << 30):
        rows = gen_rows(torch, dev, args.seed, lo, hi, d)
        torch.cuda.synchronize()
        idx = DeviceIndex.from_device_pointer(rows.data_ptr(), n_local, d, device=dev_index, row_offset=lo,
                                              dtype=args.dtype)
    else:                                                         # (configs[3]: a 12.5M-row shard, block by block)
        idx = build_index(torch, dev, dev_index, args.seed, lo, hi, d, args.dtype)
    if args.variant:
        idx.set_variant(args.variant)
    if not keep_rows_for_cpu:
        rows = None
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
    if world == 1 and n_local * d:
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
    if rank == 0 and n_local == 1_000_000 and d == 1536 and args.dtype == "f32":
        traffic, traffic_src = profile_traffic("r[0-9]_summary.json", "gemv_f32", "hbm_bytes_per_launch")

    # ---- secondary figure (N = 1 only; not `value`): batched search, 16 queries
    # share one pass over the corpus (north_star: >= 10,000 queries/s at 1 GPU)
    batched = None
    if world == 1 and args.batch and main_cfg:
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
    if world == 1 and args.concurrent > 1 and n_local * d and main_cfg:
        import threading
        gc = torch.Generator(device=dev)
        gc.manual_seed(args.seed + 4242)
        qc = torch.randn((256, d), device=dev, dtype=torch.float32, generator=gc)
        qc = (qc / qc.norm(dim=1, keepdim=True)).cpu().numpy()
        concurrent = {"callers": args.concurrent, "unit": "queries/s", "note": "Python threads, one query per call, host API"}
        # (the solo figure is opt-in: 64 uncoordinated callers put 8 score kernels on the card at once, each
        #  several times slower than alone -- that would sit in the kernel statistics of the dominant kernel that
        #  profiles/ records for this very command; alone they get `value`, ~1,140 queries/s, whatever their number)
        def run_callers(seconds):
            done = [0] * args.concurrent
            stop = time.time() + seconds

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
            return sum(done), time.time() - t0
        for mode in (("solo", "coalesced") if args.concurrent_solo else ("coalesced",)):
            idx.set_coalesce(mode == "coalesced")
            p0, a0 = idx.coalesce_stats()
            # The Python-caller figure swings between runs and boxes (BENCH_r02 13.1 k, BENCH_r03 10.8 k with the same
            # coalescer: 64 threads hand the GIL around between every call, so how many are queued when a pass forms is
            # scheduler noise -- 12 to 15 per pass).  Three runs, median reported, min / max beside it; the steady
            # figure is `c_threads` below (no GIL).
            runs = [run_callers(1.0) for _ in range(3 if mode == "coalesced" else 1)]
            rates = sorted(n_ / dt for n_, dt in runs)
            n_med, dt_med = sorted(runs, key=lambda r: r[0] / r[1])[len(runs) // 2]
            concurrent[mode] = rates[len(rates) // 2]
            concurrent[mode + "_mean_latency_ms"] = 1e3 * dt_med * args.concurrent / max(n_med, 1)
            if mode == "coalesced":
                concurrent["coalesced_runs"] = {"n": len(rates), "min": rates[0], "max": rates[-1]}
                p1, a1 = idx.coalesce_stats()
                concurrent["queries_per_corpus_pass"] = (a1 - a0) / max(p1 - p0, 1)
        idx.set_coalesce(False)
        # the same protocol from C threads (no GIL): tools/coalesce_bench.c, built by __graft_entry__.build(), run as
        # a child process on its own 1M x 1536 index (a pure-C caller of the ABI: svs_index_create + svs_index_search)
        exe = os.path.join(ROOT, "tools", "coalesce_bench_c")
        if os.path.exists(exe) and args.dtype == "f32":
            import subprocess
            try:
                r = subprocess.run([exe, str(n_local), str(d), "1.5", str(args.concurrent), "1"], capture_output=True, text=True, timeout=300, env=child_env())
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
    want = [c.strip() for c in str(args.configs).split(",") if c.strip()] if (world == 1 and main_cfg) else []
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
    if "3" in want and d == 1536:
        n3 = 12_500_000
        i3 = build_index(torch, dev, dev_index, args.seed + 3, 0, n3, d, "f16")
        configs.append(single_query_config(torch, i3, "BASELINE.json configs[3], one GPU's share: 12.5M docs x dim 1536 fp16 (of 100M over 8 GPUs), "
                                           "top-100, single query, HBM-resident GEMV + top-k", n3, d, "f16", k, args.seed + 3))
        i3.release()
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
    if world == 1 and args.kb and main_cfg:
        import subprocess
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--kb-only", "--seed", str(args.seed), "--k", str(k)],
                               capture_output=True, text=True, timeout=900, env=child_env())
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
                "workload": ("BASELINE.json configs[1]: " if (args.dtype == "f32" and n_total == 1_000_000 and d == 1536) else
                             ("BASELINE.json configs[3]: " if (args.dtype == "f16" and n_local == 12_500_000 and d == 1536 and world == 8) else
                              ("BASELINE.json configs[3] at %d of its 8 GPUs (12.5M rows per GPU): " % world
                               if (args.dtype == "f16" and n_local == 12_500_000 and d == 1536) else ""))) +
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
            "kb_retrieve_1m_p50_ms": kb_out.get("kb_retrieve_1m_p50_ms"),
            "cold_start_s": kb_out.get("cold_start_s"),
            "async_retrieve": kb_out.get("async_retrieve"),
            "kb": {k_: v for k_, v in kb_out.items() if k_ in ("kb_retrieve", "cold_start", "kb_1m")} or None,
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
    if rank == 0 and world == 1 and not args.no_cpu_baseline and main_cfg:
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
            # `cores` = the threads of the BLAS pool numpy's np.dot runs on (not torch's OpenMP pool, which the oracle never uses)
            nb = [p.get("num_threads", 1) for p in info if p.get("user_api") == "blas" and "numpy" in str(p.get("filepath", ""))] or \
                 [p.get("num_threads", 1) for p in info if p.get("user_api") == "blas"]
            thr = max(nb or [1])
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
