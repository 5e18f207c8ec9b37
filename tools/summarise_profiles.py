#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect_profiles.sh (gpurun_out/<tag>_*) into the
summaries committed under profiles/: <tag>_bench.json, <tag>_bench_kernel_stats.csv,
<tag>_pmc_hbm.json (HBM bytes per launch, corrected as MI355X_MICROARCH.md prescribes) and
<tag>_pmc_sq.json (issue-side counters of the dominant kernel).   usage: summarise_profiles.py r01c"""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
bench = json.loads(open(os.path.join(G, f"{tag}_bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(P, f"{tag}_bench.json"), "w"))
kernel = bench["roofline"]["kernel"].split(" ")[0].replace("kr::", "")
def newest(pattern):  # gpurun merges every call's output into the same directory: take the last run's file
    return sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)[-1:]
st = newest(os.path.join(G, f"{tag}_stats", "**", "*kernel_stats.csv"))
if st:
    shutil.copy(st[0], os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
    for row in csv.DictReader(open(st[0])):
        if kernel in row["Name"]:
            print("kernel stats:", row["Name"][:70], "calls", row["Calls"], "avg ns", row["AverageNs"])

def counters(sub):
    f = newest(os.path.join(G, f"{tag}_{sub}", "**", "*counter_collection.csv"))
    acc = {}
    if not f: return acc, 0
    rows = [row for row in csv.DictReader(open(f[0])) if kernel in row["Kernel_Name"]]
    if not rows: return acc, 0
    last = max(int(row["Dispatch_Id"]) for row in rows)  # the timed launch is the last one of that kernel
    for row in rows:
        if int(row["Dispatch_Id"]) == last:
            acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return acc, len({row["Dispatch_Id"] for row in rows})

fetch, nf = counters("fetch"); write, nw = counters("write")
if fetch and write:
    fk, wk = fetch["FETCH_SIZE"], write["WRITE_SIZE"]
    hbm = {"source": "rocprofv3 --kernel-trace --pmc <X> --output-format csv -- python3 bench.py --no-cpu (separate passes for FETCH_SIZE and WRITE_SIZE), tools/collect_profiles.sh",
           "kernel": bench["roofline"]["kernel"], "workload": f"B={bench['config']['rods_per_gpu']} N={bench['config']['N']} {bench['dtype']} Euler",
           "sim_path": 2 if "persistent" in bench["roofline"]["kernel"] else 1, "steps_per_launch": bench["steps"] if bench["roofline"]["launches"] == 1 else 1,
           "launches_profiled": nf, "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
           "hbm_bytes_per_launch_corrected": int(2 * fk * 1024 + wk * 1024),
           "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE on gfx950 counts 1/2 of the bytes of 16-B-per-lane reads -> doubled (upper bound here: the kernel only reads its initial state and the controls); WRITE_SIZE exact for 16-B-per-lane stores; counters are in KB"}
    json.dump(hbm, open(os.path.join(P, f"{tag}_pmc_hbm.json"), "w"), indent=1)
    print("HBM bytes/launch", hbm["hbm_bytes_per_launch_corrected"], "algorithmic", bench["roofline"]["hbm"]["algorithmic_bytes_per_launch"])
sq, ns = counters("sq")
if sq:
    sq_out = {"source": "rocprofv3 --kernel-trace --pmc SQ_* (one pass), tools/collect_profiles.sh", "kernel": bench["roofline"]["kernel"],
              "workload": f"B={bench['config']['rods_per_gpu']} N={bench['config']['N']} {bench['dtype']} Euler",
              "steps_per_launch": bench["steps"] if bench["roofline"]["launches"] == 1 else 1,
              "per_launch": sq, "launches_profiled": ns,
              "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)"}
    if sq.get("SQ_WAVE_CYCLES"):
        sq_out["valu_active_fraction_of_wave_cycles"] = sq.get("SQ_ACTIVE_INST_VALU", 0) / sq["SQ_WAVE_CYCLES"]
    json.dump(sq_out, open(os.path.join(P, f"{tag}_pmc_sq.json"), "w"), indent=1)
    print("SQ:", {k: f"{v:.3g}" for k, v in sq.items()})
