"""CPU: the parts of bench.py that decide whether the first multi-GPU run can fail for reasons unrelated to the GPUs
(VERDICT r3 item 4), and the evidence-hygiene rule of item 7: PMC traffic is quoted only from a profile of THESE kernel
sources.  No GPU is touched."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_without_gpus_fails_promptly_and_loudly():
    """`python bench.py --gpus 2` with no launcher on a box without a GPU: the two ranks it starts die on the
    "needs an MI355X" assertion; the launcher must report that and exit non-zero -- not hang in a rendezvous."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SVS_BENCH_LAUNCH_TIMEOUT"] = "120"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--rows", "1000"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        return   # (on a GPU box this command is a real run: covered by tests/test_sharded_gpu.py)
    assert r.returncode != 0
    assert "rank exit codes" in r.stderr and "MI355X" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert time.time() - t0 < 200


def test_children_start_without_the_profilers_preload(monkeypatch):
    import bench
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "x")
    monkeypatch.setenv("HSA_TOOLS_LIB", "y")
    monkeypatch.setenv("ROCPROFILER_METRICS_PATH", "z")
    env = bench.child_env({"RANK": "1"})
    assert env["RANK"] == "1" and "PATH" in env
    for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_METRICS_PATH"):
        assert k not in env


def test_traffic_is_quoted_only_from_a_profile_of_these_kernel_sources(tmp_path, monkeypatch):
    import bench
    from svs_amd.buildinfo import csrc_sha16
    here = csrc_sha16()
    (tmp_path / "profiles").mkdir()
    pmc = {"void svs::gemv_f32_oneshot_kernel<6, 1, 16, true, false, false>": {"hbm_bytes_per_launch": 6.148e9}}
    json.dump({"csrc_sha16": "0123456789abcdef", "pmc": pmc}, open(tmp_path / "profiles" / "r8_summary.json", "w"))
    json.dump({"pmc": pmc}, open(tmp_path / "profiles" / "r7_summary.json", "w"))          # unstamped (rounds 1-3)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    val, why = bench.profile_traffic("r[0-9]_summary.json", "gemv_f32", "hbm_bytes_per_launch")
    assert val is None and "other kernel sources" in why and here in why
    json.dump({"csrc_sha16": here, "pmc": pmc}, open(tmp_path / "profiles" / "r9_summary.json", "w"))
    val, why = bench.profile_traffic("r[0-9]_summary.json", "gemv_f32", "hbm_bytes_per_launch")
    assert val == 6.148e9 and "r9_summary.json" in why


def test_csrc_hash_covers_every_kernel_source():
    from svs_amd import buildinfo
    names = {os.path.basename(p) for p in buildinfo.csrc_files()}
    on_disk = {f for f in os.listdir(os.path.join(ROOT, "svs_amd", "csrc")) if f.endswith((".h", ".hip"))}
    assert on_disk <= names and "svs_amd.h" in names and "Makefile" in names
    assert len(buildinfo.csrc_sha16()) == 16
