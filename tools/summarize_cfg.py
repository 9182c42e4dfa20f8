#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>_<name>_{trace,pmc,fetch} (tools/profile_round.sh) into
profiles/<tag>_<name>_kernel_stats.csv and profiles/<tag>_<name>_summary.json.

Derived figures (per kernel, averages over its launches in the counter pass):
  gfx_clock_ghz  = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration   (the counter sums the XCDs)
  MfmaUtil_pct   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
  lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  hbm_read_bytes = 2 * FETCH_SIZE KiB   (gfx950 reports half of a wide coalesced stream)
usage: summarize_cfg.py TAG NAME "command line that was profiled"
"""
import collections, csv, glob, json, os, shutil, sys

tag, name = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def newest(pat):
    """The profiled program's own file of the most recent run: gpurun merges every run's files
    into the same directory (one set per process id), so take the files written with the
    latest merge and, among those, the largest (helper processes leave near-empty ones)."""
    f = glob.glob(pat)
    if not f:
        return None
    t = max(os.path.getmtime(x) for x in f)
    return max((x for x in f if os.path.getmtime(x) >= t - 120), key=os.path.getsize)


short = lambda k: k.split("(")[0]
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

out = {"tag": tag, "git_head": git_head(), "csrc_sha16": csrc_stamp(), "config": name, "command": cmd, "kernels": {}, "pmc": {}}
stats = newest(os.path.join(src, f"prof_{tag}_{name}_trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats, os.path.join(dst, f"{tag}_{name}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats)):
        if "svs::" in r["Name"]:
            out["kernels"][short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
for kind in ("pmc", "fetch"):
    cc = newest(os.path.join(src, f"prof_{tag}_{name}_{kind}", "*", "*_counter_collection.csv"))
    if not cc:
        continue
    tr = cc.replace("_counter_collection.csv", "_kernel_trace.csv")
    dur = collections.defaultdict(list)
    if os.path.exists(tr):
        for r in csv.DictReader(open(tr)):
            if "svs::" in r["Kernel_Name"]:
                dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        if "svs::" in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        p = out["pmc"].setdefault(k, {})
        for c, x in v.items():
            p[c] = sum(x) / len(x)
        if k in dur:
            p[f"duration_us_in_{kind}_pass"] = sum(dur[k]) / len(dur[k]) / 1e3
        if "GRBM_GUI_ACTIVE" in p and k in dur:
            p["gfx_clock_ghz"] = p["GRBM_GUI_ACTIVE"] / 8 / (sum(dur[k]) / len(dur[k]))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in p and p.get("GRBM_GUI_ACTIVE"):
            p["MfmaUtil_pct"] = 100.0 * p["SQ_VALU_MFMA_BUSY_CYCLES"] / (p["GRBM_GUI_ACTIVE"] / 8 * 1024)
        if p.get("SQ_LDS_IDX_ACTIVE"):
            p["lds_bank_conflict_frac"] = p.get("SQ_LDS_BANK_CONFLICT", 0.0) / p["SQ_LDS_IDX_ACTIVE"]
        if "FETCH_SIZE" in p:
            p["hbm_read_bytes_per_launch"] = 2.0 * p["FETCH_SIZE"] * 1024
json.dump(out, open(os.path.join(dst, f"{tag}_{name}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
