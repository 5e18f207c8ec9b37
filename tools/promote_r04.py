#!/usr/bin/env python3
"""Build container: copy what tools/collect_r04.sh left under gpurun_out/<tag>_* into profiles/ (small summaries only).
    python tools/promote_r04.py r02a"""
import csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarise_profiles.py"), tag], check=False)
def newest(pattern):
    f = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return f[-1] if f else None
def stats_rows(path, keep):
    rows = []
    for row in csv.DictReader(open(path)):
        if any(k in row["Name"] for k in keep):
            rows.append({"kernel": row["Name"][:110], "calls": int(row["Calls"]), "avg_us": round(float(row["AverageNs"]) / 1e3, 2),
                         "total_ms": round(float(row["TotalDurationNs"]) / 1e6, 3)})
    return rows
def counters(dirname, kernel_substr):
    f = newest(os.path.join(G, dirname, "**", "*counter_collection.csv"))
    if not f: return None
    acc, n = {}, set()
    for row in csv.DictReader(open(f)):
        if kernel_substr in row["Kernel_Name"]:
            acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"]); n.add(row["Dispatch_Id"])
    if not acc: return None
    rec = {"kernel": kernel_substr, "dispatches": len(n), "sum_over_dispatches": acc}
    wc = acc.get("SQ_WAVE_CYCLES")
    if wc:
        # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md):
        # per wave-cycle figures; with w waves resident per SIMD the SIMD-level matrix-pipe utilisation is w times the first one
        rec["per_wave_cycle"] = {"mfma_busy": round(acc.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc), 4),
                                 "valu_active": round(acc.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4),
                                 "wait_any": round(acc.get("SQ_WAIT_ANY", 0) / wc, 4),
                                 "lds_bank_conflict": round(acc.get("SQ_LDS_BANK_CONFLICT", 0) / (4 * wc), 4)}
    return rec
out = {}
for p in ("f64", "f32"):
    st = newest(os.path.join(G, f"{tag}_nn_{p}_stats", "**", "*kernel_stats.csv"))
    rec = {"what": f"tools/simnn_only.py {p} 40: B=1024, N=100, 28-64-64-25 ELU inside every sweep, 40 steps from the straight rod"}
    if st: rec["kernel_stats"] = stats_rows(st, ("ms_sim_kernel", "ms_step_kernel"))
    log = os.path.join(G, f"{tag}_nn_{p}.log")
    if os.path.exists(log): rec["line"] = [l.strip() for l in open(log) if "ms/step" in l][-1:]
    c = counters(f"{tag}_nn_{p}_sq", "ms_sim_kernel")
    if c:
        c["waves_per_simd"] = 1
        rec["sq"] = c
    out[f"simulate_nn_{p}"] = rec
for cfg in ("cfg3", "cfg4"):
    st = os.path.join(G, f"{tag}_train_{cfg}_stats.csv")
    rec = {"what": f"tools/train_only.py {cfg}: 20 training epochs through kr_train_epoch (forward + loss, backward pass(es), one tail launch: slab and loss sums, Adam + clamp + plateau schedule, fragment update)"}
    if os.path.exists(st): rec["kernel_stats"] = stats_rows(st, ("mlp_", "loss", "adam", "pack_", "reduce_", "tail"))
    for k in ("mlp_bwd3a", "mlp_bwd3b", "mlp_bwd2", "mlp_fwd"):
        c = counters(f"{tag}_train_{cfg}_sq", k)
        if c:
            c["waves_per_simd"] = 2
            rec.setdefault("sq", []).append(c)
    out[f"train_{cfg}"] = rec
st = newest(os.path.join(G, f"{tag}_ode_stats", "**", "*kernel_stats.csv"))
rec = {"what": "tools/aux_bench.py ode: kr_ode_batch, 78 sizeof(T) algorithmic bytes per row"}
if st: rec["kernel_stats"] = stats_rows(st, ("ode_batch",))
t = os.path.join(G, f"{tag}_ode.txt")
if os.path.exists(t): rec["lines"] = [l.strip() for l in open(t) if l.startswith("ode_batch")]
out["ode_batch"] = rec
# cfg5 on the several-wavefront step kernel: kernel stats, HBM bytes per launch (FETCH_SIZE / WRITE_SIZE: units of 32 B on
# gfx950 as in tools/summarise_profiles.py) against the algorithmic (75 N + 16) s B, SQ issue figures
st = newest(os.path.join(G, f"{tag}_cfg5_stats", "**", "*kernel_stats.csv"))
rec = {"what": "tools/cfg5_only.py = bench.py extra.cfg5: B=512, N=400, fp64, 30 untimed + 60 timed steps, all steps of a call in one persistent launch (msw_sim_kernel, history records in LDS, newest states from HBM), two wavefronts per rod"}
if st: rec["kernel_stats"] = stats_rows(st, ("msw_sim_kernel", "msw_step_kernel", "ms_step_kernel"))
log = os.path.join(G, f"{tag}_cfg5.log")
if os.path.exists(log): rec["line"] = [l.strip() for l in open(log) if l.startswith("cfg5")][-1:]
tr = newest(os.path.join(G, f"{tag}_cfg5_stats", "**", "*kernel_trace.csv"))
if tr:  # the timed dispatch is the last one of the step kernel: 60 steps
    rows = [r_ for r_ in csv.DictReader(open(tr)) if "msw_sim_kernel" in r_["Kernel_Name"]]
    if rows:
        last = max(rows, key=lambda r_: int(r_["Dispatch_Id"]))
        dur = (int(last["End_Timestamp"]) - int(last["Start_Timestamp"])) * 1e-3
        rec["timed_dispatch"] = {"kernel": last["Kernel_Name"][:80], "us": round(dur, 1), "steps": 60, "us_per_step": round(dur / 60, 2),
                                 "dispatches_of_this_kernel_in_the_trace": len(rows)}
for cname, d in (("FETCH_SIZE", f"{tag}_cfg5_fetch"), ("WRITE_SIZE", f"{tag}_cfg5_write")):
    c = counters(d, "msw_sim_kernel")
    if c:
        rec.setdefault("hbm", {})[cname + "_raw_per_launch"] = c["sum_over_dispatches"].get(cname, 0.0) / max(c["dispatches"], 1)
if "hbm" in rec:
    # counters are in KB; FETCH_SIZE on gfx950 counts half the bytes of 16-B-per-lane reads -> doubled (as in
    # tools/summarise_profiles.py, MI355X_MICROARCH.md HBM section); WRITE_SIZE exact for 16-B-per-lane stores
    f_, w_ = rec["hbm"].get("FETCH_SIZE_raw_per_launch", 0.0), rec["hbm"].get("WRITE_SIZE_raw_per_launch", 0.0)
    rec["hbm"]["hbm_bytes_per_launch_corrected"] = int(2 * f_ * 1024 + w_ * 1024)
    # one persistent launch = the 60 timed steps of 512 rods on a 3-slot ring (tips out, tensions in); the full-trajectory
    # mode would write (25 N + 4) s per rod-step
    rec["hbm"]["algorithmic_bytes_per_launch"] = {"tip_only_ring": 7 * 8 * 512 * 60, "full_trajectory": (25 * 400 + 4) * 8 * 512 * 60}
c = counters(f"{tag}_cfg5_sq", "msw_sim_kernel")
if c:
    c["waves_per_simd"] = 1
    a = c["sum_over_dispatches"]
    if a.get("SQ_BUSY_CYCLES") and a.get("SQ_INSTS_VALU"):
        c["note"] = "valu issue fraction = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x launch cycles): see bench roofline for the formula"
    rec["sq"] = c
out["cfg5_msw"] = rec
json.dump(out, open(os.path.join(P, f"{tag}_kernels.json"), "w"), indent=1)
for name in ("configs.txt", "train.txt"):
    src = os.path.join(G, f"{tag}_{name}")
    if os.path.exists(src):
        open(os.path.join(P, f"{tag}_{name}"), "w").write("".join(l for l in open(src) if "amdgpu.ids" not in l))
print("profiles/ now holds:", sorted(f for f in os.listdir(P) if f.startswith(tag)))
