#!/usr/bin/env python3
"""Condense gpurun_out/{prof,pmc_*} (rocprofv3 csv) into small summaries that are committed under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
os.makedirs(os.path.join(R, "profiles"), exist_ok=True)
out = {}
def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: keep only the most recent match."""
    files = sorted(glob.glob(os.path.join(R, pattern)), key=os.path.getmtime)
    return files[-1:]


for f in newest("gpurun_out/prof/*/*kernel_stats.csv"):
    rows = list(csv.reader(open(f)))
    with open(os.path.join(R, "profiles", tag + "_bench_kernel_stats.csv"), "w") as o:
        w = csv.writer(o)
        for r in rows:
            r[0] = r[0][:96]
            w.writerow(r)
    total_calls = 0
    for r in rows[1:]:
        total_calls += int(r[1])
        if "ddqn_se_inner" in r[0]:
            out["inner_kernel_calls"] = int(r[1])
            out["inner_kernel_avg_ms"] = float(r[3]) / 1e6
    if out.get("inner_kernel_calls"):
        # every dispatch rocprofv3 saw (kernels, fills, device copies) per generation, start-up launches included
        out["launches_per_generation"] = total_calls / out["inner_kernel_calls"]
    print("".join(",".join(r[:5]) + "\n" for r in rows[:6]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in newest("gpurun_out/pmc_%s/*/*counter_collection.csv" % c):
        per_dispatch = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "ddqn_se_inner" in r["Kernel_Name"] and r["Counter_Name"] == c:
                per_dispatch[r["Dispatch_Id"]] += float(r["Counter_Value"])
        vals += list(per_dispatch.values())
    if vals:
        out[c + "_KB_per_launch"] = sum(vals) / len(vals)
if "FETCH_SIZE_KB_per_launch" in out and "WRITE_SIZE_KB_per_launch" in out:
    # MI355X_MICROARCH.md §HBM: counters are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced
    # streaming reads -> double the read side (upper bound for this kernel's 16-byte row gathers); WRITE_SIZE is exact.
    out["hbm_traffic_bytes_per_launch"] = (2.0 * out["FETCH_SIZE_KB_per_launch"] + out["WRITE_SIZE_KB_per_launch"]) * 1024.0
    out["traffic_note"] = "2*FETCH_SIZE + WRITE_SIZE (KB->bytes), per fused-kernel launch, gfx950 read-side correction applied"
# the other configurations (tools/bench_configs.py under rocprofv3 --kernel-trace --stats)
for f in newest("gpurun_out/prof_configs/*/*kernel_stats.csv"):
    rows = list(csv.reader(open(f)))
    with open(os.path.join(R, "profiles", tag + "_configs_kernel_stats.csv"), "w") as o:
        w = csv.writer(o)
        for r in rows:
            r[0] = r[0][:96]
            w.writerow(r)
cfgs = os.path.join(R, "gpurun_out", "bench_configs.jsonl")
if os.path.exists(cfgs):
    out["configs"] = [json.loads(l) for l in open(cfgs) if l.startswith("{")]
bench = os.path.join(R, "gpurun_out", "bench.json")
if os.path.exists(bench):
    try:
        out["bench"] = json.loads(open(bench).read().strip().splitlines()[-1])
    except Exception as e:  # noqa
        out["bench_error"] = str(e)
json.dump(out, open(os.path.join(R, "profiles", tag + "_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench"}, indent=1))
if "bench" in out:
    b = out["bench"]
    print("bench value %.1f %s, ms/step %.2f, kernel_ms %.2f, cpu %s" % (b["value"], b["unit"], b["ms_per_step"],
          b["roofline"]["kernel_ms"], b.get("cpu_baseline", {}).get("value")))
