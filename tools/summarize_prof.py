#!/usr/bin/env python3
"""Condenses rocprofv3 outputs (gpurun_out/prof_<tag>_{trace,fetch,write}) into
profiles/<tag>_*.  FETCH_SIZE on gfx950 reports exactly half the bytes of a wide
coalesced stream (MI355X_MICROARCH.md, HBM section) -> doubled; unit is KiB.
WRITE_SIZE is exact (KiB)."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

def newest(pat):
    # latest merge's files; among those the largest (helper processes leave near-empty ones)
    f = glob.glob(pat)
    t = max(os.path.getmtime(x) for x in f)
    return max((x for x in f if os.path.getmtime(x) >= t - 120), key=os.path.getsize)

stats = newest(os.path.join(src, f"prof_{tag}_trace", "*", "*_kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
kern = {}
for r in csv.DictReader(open(stats)):
    if "svs::" in r["Name"]:
        kern[r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                          "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                          "pct": float(r["Percentage"])}
pmc = collections.defaultdict(dict)
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = newest(os.path.join(src, f"prof_{tag}_{name}", "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "svs::" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k][ctr + "_KiB_avg"] = sum(v) / len(v)
        pmc[k]["launches_" + name] = len(v)
for k, v in pmc.items():
    rd = 2.0 * v.get("FETCH_SIZE_KiB_avg", 0.0) * 1024   # gfx950 correction: x2
    wr = v.get("WRITE_SIZE_KiB_avg", 0.0) * 1024
    v["hbm_read_bytes_per_launch"] = rd
    v["hbm_write_bytes_per_launch"] = wr
    v["hbm_bytes_per_launch"] = rd + wr
def csrc_stamp():
    """csrc_sha16 the passes were stamped with ON THE GPU BOX at profile time (tools/profile_round.sh); a profile without the
    stamp is marked so -- bench.py then refuses to quote its traffic."""
    f = os.path.join(src, f"prof_{tag}_csrc_sha.txt")
    return open(f).read().strip() if os.path.exists(f) else "unstamped"
def git_head():
    """The commit the profiled tree was at (the GPU box has no .git: summarise right after the run, before the next commit)."""
    import subprocess
    try:
        h = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "svs_amd", "bench.py", "tools/prof_batch.py"], capture_output=True, text=True).stdout.strip()
        return h + ("+uncommitted" if dirty else "")
    except Exception:
        return os.environ.get("SVS_GIT_HEAD", "unrecorded")
out = {"tag": tag, "git_head": git_head(), "csrc_sha16": csrc_stamp(), "kernels": kern, "pmc": pmc,
       "note": "FETCH_SIZE x2 (gfx950 reports half of a wide coalesced read), KiB units; WRITE_SIZE exact"}
json.dump(out, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
bench = os.path.join(src, f"prof_{tag}_trace.json")
if os.path.exists(bench):
    shutil.copy(bench, os.path.join(dst, f"{tag}_bench_line_under_rocprof.json"))
print(json.dumps(out, indent=1))
