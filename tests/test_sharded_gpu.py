"""GPU: the N > 1 pipelined device exchange with REAL ranks (two and three processes sharing the test
box's one card over gloo): what tools/rehearse_sharded.sh did by hand in round 2.  Every rank holds its
row block in HBM (row_offset = shard_bounds()[0]), single-query searches alternate between two HIP
streams, `gather_every` records share one all-gather, rank 0 copies each gathered chunk home and merges
(svs_amd/sharded.py: ShardedIndex.open / enqueue / collect -- the code bench.py --gpus N runs).  The
merged answer must be IDENTICAL, rows and score bits, to one index over the whole corpus: a row's
score does not depend on where it lives (reference analogue: np.dot over the whole matrix,
src/svs/kb.py:1623, then get_top_k, src/svs/util.py:190-203)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from synth import corpus_and_query

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n,d,nq,k,gather_every,dtype", [
    (2, 300_000, 384, 21, 100, 8, "f32"),    # 21 searches = two whole chunks of 8 + a partly filled last one
    (3, 100_003, 256, 9, 50, 4, "f16"),      # uneven shards (ceil split), three ranks, half-precision corpus
    (2, 40, 64, 5, 100, 8, "f32"),           # k > rows: every shard answers with fewer than k, padded records
])
def test_pipelined_exchange_with_real_ranks(gpu, tmp_path, world, n, d, nq, k, gather_every, dtype):
    from svs_amd import DeviceIndex
    port = _free_port()
    out = str(tmp_path / "rank0.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_gpu_worker.py"), str(r), str(world), str(port), out,
                               str(n), str(d), str(nq), str(k), str(gather_every), dtype],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    got = np.load(out)
    m, qs = corpus_and_query("gaussian", 2024, n, d, nq)
    whole = DeviceIndex(m, dtype=dtype)
    count = min(k, n)
    assert got["rows"].shape == (nq, count)
    for i in range(nq):
        exp = whole.search(qs[i], k)
        assert [int(x) for x in got["rows"][i]] == [r for _, r in exp], f"query {i}: rows differ from one index"
        assert [float(x) for x in got["scores"][i]] == [s for s, _ in exp], f"query {i}: score bits differ from one index"
    whole.release()


ROOT = os.path.dirname(HERE)


def _bench_line(args, env_extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)                       # no launcher: bench.py must start its own ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    import json
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks(gpu):
    """`python bench.py --gpus 2` with NO launcher and no RANK in the environment (the shape of the driver's N = 1
    command): bench.py starts two rank processes of itself, they rendezvous on 127.0.0.1 (gloo here: the two ranks
    share the box's one card), rank 0's JSON line comes back alone, and with SVS_BENCH_VERIFY=1 the merged results of
    the timed steps equal one index over the whole corpus, rows and score bits."""
    line = _bench_line(["--gpus", "2", "--steps", "20", "--warmup", "5", "--rows", "200000"],
                       {"SVS_BENCH_BACKEND": "gloo", "SVS_BENCH_VERIFY": "1"})
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["scaling"] == "strong"
    assert line["sharded_check"] == {"queries": 16, "mismatches": 0}
    assert line["config"]["rows_per_gpu"] == 100000 and line["value"] > 0


def test_bench_config3_weak_and_multi_modes(gpu):
    """`--config 3` (BASELINE.json configs[3]: f16 rows per GPU, weak scaling; here with a reduced --rows so that two
    ranks and the whole-corpus check fit the time budget) through self-launched ranks, and `--multi`: the same shards
    in one process through svs_multi_search."""
    line = _bench_line(["--gpus", "2", "--config", "3", "--rows", "150000", "--steps", "10", "--warmup", "3"],
                       {"SVS_BENCH_BACKEND": "gloo", "SVS_BENCH_VERIFY": "1"})
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["dtype"] == "f16"
    assert line["config"]["rows_per_gpu"] == 150000 and line["sharded_check"]["mismatches"] == 0
    multi = _bench_line(["--gpus", "2", "--multi", "--rows", "200000", "--steps", "10", "--warmup", "3"], {})
    assert multi["n_gpus"] == 2 and "svs_multi" in multi["config"]["path"] and multi["config"]["rows_per_shard"] == [100000, 100000]
    assert multi["value"] > 0


def test_bench_rccl_exchange_path_with_the_one_rank_a_box_has(gpu):
    """SVS_BENCH_FORCE_DIST=1: the N > 1 code path over the REAL backend (RCCL process group, records in HBM, the
    all-gather + copy home on the exchange stream, chunk-by-chunk host merge on 64-bit keys) with one rank -- 29 steps,
    so that the warm-up's partly filled chunk is re-sent and the last chunk is partial -- and every timed result checked
    against the numpy oracle by bench.py's own spot check (the CPU baseline leg, 8 queries)."""
    line = _bench_line(["--rows", "200000", "--steps", "29", "--warmup", "5", "--batch", "", "--concurrent", "0", "--configs", "",
                        "--kb", "0", "--cpu-iters", "3"], {"SVS_BENCH_FORCE_DIST": "1"})
    assert line["n_gpus"] == 1 and line["steps"] == 29 and line["config"]["steps_per_exchange"] == 8
    assert line["config"]["searches_in_flight"] >= 2
    assert line["parity_spot_check"] == {"queries": 8, "mismatches": 0}
    assert line["value"] > 0
