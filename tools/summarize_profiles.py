#!/usr/bin/env python3
"""Condense gpurun_out/{prof*,pmc*} (rocprofv3 csv, written by tools/gpu_round.sh) into the small files committed under profiles/:
<tag>_bench_kernel_stats.csv, <tag>_cfgN_kernel_stats.csv (one per BASELINE configuration), <tag>_pmc_sq_<kernel>.txt, <tag>_summary.json."""
import collections
import csv
import glob
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(R, "profiles")
os.makedirs(P, exist_ok=True)
out = {}


def newest(pattern):
    files = sorted(glob.glob(os.path.join(R, pattern)), key=os.path.getmtime)
    return files[-1:]


def copy_stats(pattern, dest, kernel_sub):
    res = {}
    for f in newest(pattern):
        rows = list(csv.reader(open(f)))
        with open(os.path.join(P, dest), "w") as o:
            w = csv.writer(o)
            for r in rows:
                r[0] = r[0][:96]
                w.writerow(r)
        total = sum(int(r[1]) for r in rows[1:])
        for r in rows[1:]:
            if kernel_sub in r[0]:
                res = {"kernel": r[0][:64], "calls": int(r[1]), "avg_ms": float(r[3]) / 1e6, "min_ms": float(r[5]) / 1e6 if len(r) > 6 else None, "max_ms": float(r[6]) / 1e6 if len(r) > 6 else None,
                       "dispatches_per_generation": total / max(1, int(r[1]))}
    return res


def pmc(pattern, kernel_sub, counter):
    vals = []
    for f in newest(pattern):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
        vals += list(per.values())
    return sum(vals) / len(vals) if vals else None


def traffic(prefix, kernel_sub):
    f, w = pmc(prefix + "FETCH_SIZE/*/*counter_collection.csv", kernel_sub, "FETCH_SIZE"), pmc(prefix + "WRITE_SIZE/*/*counter_collection.csv", kernel_sub, "WRITE_SIZE")
    if f is None or w is None:
        return None
    # MI355X_MICROARCH.md (HBM): counters in KB; gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> x2 on the read side
    return {"FETCH_SIZE_KB_per_launch": f, "WRITE_SIZE_KB_per_launch": w, "hbm_traffic_bytes_per_launch": (2.0 * f + w) * 1024.0,
            "note": "2*FETCH_SIZE + WRITE_SIZE (KB -> bytes) per launch, gfx950 read-side correction"}


def steady_state_dispatches(pattern, kernel_sub):
    """Dispatches (kernels, fills, device copies) between the last two launches of the fused kernel in the kernel trace: one
    steady-state generation, start-up launches excluded."""
    for f in newest(pattern):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        idx = [i for i, r in enumerate(rows) if kernel_sub in r["Kernel_Name"]]
        if len(idx) >= 2:
            return idx[-1] - idx[-2], [rows[i]["Kernel_Name"][:48] for i in range(idx[-2] + 1, idx[-1] + 1)]
    return None, None


head = copy_stats("gpurun_out/prof/*/*kernel_stats.csv", tag + "_bench_kernel_stats.csv", "ddqn_se_inner")
if head:
    out["inner_kernel_calls"], out["inner_kernel_avg_ms"] = head["calls"], head["avg_ms"]
    n, names = steady_state_dispatches("gpurun_out/prof/*/*kernel_trace.csv", "ddqn_se_inner")
    out["launches_per_generation"] = n if n is not None else head["dispatches_per_generation"]
    out["launches_of_a_generation"] = names
t = traffic("gpurun_out/pmc_", "ddqn_se_inner")
if t:
    out.update({k: v for k, v in t.items() if k != "note"})
    out["traffic_note"] = t["note"]
kern = {1: "ddqn_se_inner", 2: "dueling_wavechain", 3: "ql_rn_inner", 4: "td3_wavechain"}
out["configs"] = {}
for n in (1, 2, 3, 4):
    rec = copy_stats("gpurun_out/prof_cfg%d/*/*kernel_stats.csv" % n, ("%s_cfg%d_kernel_stats.csv" % (tag, n + 1)) if n > 1 else (tag + "_cfg2_strong_shard_kernel_stats.csv"), kern[n])
    try:
        line = [l for l in open(os.path.join(R, "gpurun_out", "prof_cfg%d.json" % n)) if l.startswith("[")][-1]
        rec["bench_record_under_rocprof"] = json.loads(line)[0]
    except Exception as e:  # noqa
        rec["bench_record_error"] = str(e)
    tr = traffic("gpurun_out/pmccfg_%d_" % n, kern[n])
    if tr:
        rec["traffic"] = tr
        # fabric traffic per LEARN step against the algorithmic bytes of a learn step (4 * [B * row + 8 * P_agent]); the
        # launch's learn steps come from the bench record taken under rocprofv3 (steps = timed generations, the PMC value is per launch)
        br = rec.get("bench_record_under_rocprof") or {}
        if br.get("learn_steps") and br.get("steps"):
            learn_per_launch = float(br["learn_steps"])      # (the record's counters are those of ONE generation = one launch)
            alg = {2: 4 * (128 * 15 + 8 * 67460), 4: 4 * (192 * 42 + 8 * 59016)}.get(n)
            rec["traffic_per_learn_step_bytes"] = tr["hbm_traffic_bytes_per_launch"] / learn_per_launch
            if alg:
                rec["algorithmic_bytes_per_learn_step"] = alg
                rec["traffic_over_algorithmic_per_learn_step"] = rec["traffic_per_learn_step_bytes"] / alg
    out["configs"]["BASELINE configs[%d]" % n if n > 1 else "BASELINE configs[1] strong-scaling shard (8 workers = 24 chains, teams)"] = rec


def sq(pattern, kernel_sub, dest):
    lines = []
    for f in sorted(glob.glob(os.path.join(R, pattern))):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        disp = len({r["Dispatch_Id"] for r in csv.DictReader(open(f)) if kernel_sub in r["Kernel_Name"]})
        for k in acc:
            lines.append("%s %.6g per launch (%d launches)" % (k, acc[k] / max(disp, 1), disp))
    if lines:
        open(os.path.join(P, dest), "w").write("\n".join(lines) + "\n")
    return lines


sq("gpurun_out/pmc_sq_*/*/*counter_collection.csv", "ddqn_se_inner", tag + "_pmc_sq_ddqn_se_inner_kernel.txt")
sq("gpurun_out/pmccfg_2_sq_*/*/*counter_collection.csv", "dueling_wavechain", tag + "_pmc_sq_dueling_wavechain_kernel.txt")
sq("gpurun_out/pmccfg_4_sq_*/*/*counter_collection.csv", "td3_wavechain", tag + "_pmc_sq_td3_wavechain_kernel.txt")
r1 = os.path.join(R, "gpurun_out", "bench_rccl1.json")
if os.path.exists(r1):
    try:
        d1 = json.loads(open(r1).read().strip().splitlines()[-1])
        out["bench_one_rank_rccl"] = {k: d1.get(k) for k in ("value", "ms_per_step", "ranks", "n_gpus")}
        out["bench_one_rank_rccl"]["graphs_per_generation"] = d1.get("config", {}).get("graphs_per_generation")
    except Exception as e:  # noqa
        out["bench_one_rank_rccl_error"] = str(e)
bench = os.path.join(R, "gpurun_out", "bench.json")
if os.path.exists(bench):
    try:
        out["bench"] = json.loads(open(bench).read().strip().splitlines()[-1])
    except Exception as e:  # noqa
        out["bench_error"] = str(e)
json.dump(out, open(os.path.join(P, tag + "_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "bench"}, indent=1)[:4000])
