#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect.sh (gpurun_out/<tag>_<section>_<pass>/) into the small summaries that are
committed under profiles/ - ONE script for all rounds (it replaces summarise_profiles.py and promote_r02/r03/r04.py).

    python tools/profile_summary.py <tag>

What protects the numbers (a counter file of another program was summarised under the headline's
name): a counter row is only ever attributed to a workload if
  (1) its Kernel_Name holds the kernel's base name AND the instantiation's arithmetic type (`<double` / `<float`), and
  (2) the dispatch's duration (End - Start of the same row) agrees within 20 % with the HIP-event time the SAME process
      printed for that launch (every pass keeps the stdout of the program it profiled), and
  (3) for the bench passes, that process's ms_per_step agrees within 20 % with <tag>_bench.json.
Anything else raises ProfileMismatch; nothing is written for that section.  tests/test_host_cpu.py feeds this module a
mislabelled synthetic CSV and expects the refusal.

Written to profiles/: <tag>_bench.json, <tag>_bench_steps20.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_hbm.json,
<tag>_pmc_sq.json, <tag>_pmc_ops.json (executed arithmetic per leg), <tag>_kernels.json, <tag>_train.txt, <tag>_configs.txt."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 0.20
MOPS_FLOPS = 512  # SQ_INSTS_VALU_MFMA_MOPS_*: matrix operations (add or mul) / 512, full EXEC mask assumed (rocprofv3 -L)
PEAK_TF = {"F64": 78.6, "F32": 157.3, "BF16": 2516.6}  # dense MFMA peaks (fp64: AMD public figure; others MI355X_MICROARCH.md)


class ProfileMismatch(Exception):
    pass


def kernel_base(label):
    """'kr::mso_sim_kernel (persistent, overlapped steps)' -> 'mso_sim_kernel'"""
    return label.split(" ")[0].replace("kr::", "")


def dtype_token(dtype):
    return "<double" if dtype in ("f64", "double") else "<float"


def read_rows(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def group_dispatches(rows, base, dtok, name_key="Kernel_Name"):
    """{dispatch id: {"name", "ns", "counters": {name: summed value}}} of the rows whose kernel name holds `base` and `dtok`
    (counter rows come one per counter, XCC and dimension: values of one dispatch and one counter are summed)."""
    out = {}
    for r in rows:
        n = r[name_key]
        if base not in n or (dtok and dtok not in n):
            continue
        d = out.setdefault(int(r["Dispatch_Id"]), {"name": n, "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "counters": {}})
        if "Counter_Name" in r:
            d["counters"][r["Counter_Name"]] = d["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return out


def timed_dispatches(rows, base, dtype, expect_us, what, tol=TOL):
    """The dispatches of kernel `base` in arithmetic type `dtype` whose duration is within `tol` of expect_us (the HIP-event
    time the profiled process itself reported for its timed launch).  Raises ProfileMismatch when the instantiation does
    not occur at all or none of its dispatches has the expected duration."""
    dtok = dtype_token(dtype)
    any_type = group_dispatches(rows, base, "")
    if not any_type:
        raise ProfileMismatch(f"{what}: no dispatch of {base} in this file")
    typed = group_dispatches(rows, base, dtok)
    if not typed:
        seen = sorted({d["name"][:60] for d in any_type.values()})
        raise ProfileMismatch(f"{what}: {base} occurs only in other instantiations than {dtok}...>: {seen} - this file was "
                              f"written by another workload")
    good = {k: d for k, d in typed.items() if abs(d["ns"] * 1e-3 - expect_us) <= tol * expect_us}
    if not good:
        durs = sorted(round(d["ns"] * 1e-3, 1) for d in typed.values())
        raise ProfileMismatch(f"{what}: no dispatch of {base}{dtok}...> lasts {expect_us:.1f} us +- {int(tol * 100)} % "
                              f"(durations seen: {durs[:12]} us) - this file was written by another workload")
    return good


def mean_counters(disp):
    acc = {}
    for d in disp.values():
        for k, v in d["counters"].items():
            acc[k] = acc.get(k, 0.0) + v
    return {k: v / len(disp) for k, v in acc.items()}


def newest(pattern):
    f = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


def last_json_line(path, prefix=""):
    if not os.path.exists(path):
        return None
    for line in reversed(open(path).read().strip().splitlines()):
        line = line.strip()
        if prefix and line.startswith(prefix):
            line = line[len(prefix):].strip()
        elif prefix:
            continue
        if line.startswith("{"):
            try:
                return json.loads(line)
            except json.JSONDecodeError:
                continue
    return None


def executed_flops(c, lanes=64):
    """Executed arithmetic of one launch from the SQ op counters: vector flops assume all `lanes` of an instruction are
    active (an UPPER bound: the step kernels run 58 of 64 lanes), FMA = 2 flops, a transcendental = 1."""
    v64 = lanes * (c.get("SQ_INSTS_VALU_ADD_F64", 0) + c.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * c.get("SQ_INSTS_VALU_FMA_F64", 0)
                   + c.get("SQ_INSTS_VALU_TRANS_F64", 0))
    v32 = lanes * (c.get("SQ_INSTS_VALU_ADD_F32", 0) + c.get("SQ_INSTS_VALU_MUL_F32", 0) + 2 * c.get("SQ_INSTS_VALU_FMA_F32", 0)
                   + c.get("SQ_INSTS_VALU_TRANS_F32", 0))
    m = {t: MOPS_FLOPS * c.get(f"SQ_INSTS_VALU_MFMA_MOPS_{t}", 0) for t in ("F64", "F32", "BF16")}
    return {"valu_f64": v64, "valu_f32": v32, "mfma_f64": m["F64"], "mfma_f32": m["F32"], "mfma_bf16": m["BF16"]}


# ---------------------------------------------------------------------------------------------------------------------
def summarise(tag, G, P, log=print):
    os.makedirs(P, exist_ok=True)
    problems = []

    def section(fn):
        try:
            fn()
        except ProfileMismatch as e:
            problems.append(str(e))
            log("REFUSED: " + str(e))
        except FileNotFoundError as e:
            log(f"(skipped: {e})")

    bench = last_json_line(os.path.join(G, f"{tag}_bench.json"))
    if bench is None:
        raise SystemExit(f"no bench line under {G}/{tag}_bench.json")
    json.dump(bench, open(os.path.join(P, f"{tag}_bench.json"), "w"))
    b20 = last_json_line(os.path.join(G, f"{tag}_bench_steps20.json"))
    if b20:
        json.dump(b20, open(os.path.join(P, f"{tag}_bench_steps20.json"), "w"))
    base = kernel_base(bench["roofline"]["kernel"])
    dtype = bench["dtype"]
    B, N = bench["config"]["rods_per_gpu"], bench["config"]["N"]
    workload = f"B={B} N={N} {dtype} Euler"

    def bench_pass(name):
        """(rows, expected us per timed dispatch, steps per launch) of one pass over `bench.py --no-cpu --no-extra`."""
        d = os.path.join(G, f"{tag}_{name}")
        own = last_json_line(d + ".out")
        if own is None:
            raise FileNotFoundError(f"{d}.out holds no bench line")
        if own["dtype"] != dtype or kernel_base(own["roofline"]["kernel"]) != base:
            raise ProfileMismatch(f"{name}: the profiled process ran {own['roofline']['kernel']} / {own['dtype']}, the bench line "
                                  f"{bench['roofline']['kernel']} / {dtype}")
        if abs(own["ms_per_step"] - bench["ms_per_step"]) > TOL * bench["ms_per_step"]:
            raise ProfileMismatch(f"{name}: {own['ms_per_step']} ms per step under the profiler against {bench['ms_per_step']} in "
                                  f"{tag}_bench.json (> {int(TOL * 100)} %)")
        f = newest(os.path.join(d, "**", "*counter_collection.csv")) or newest(os.path.join(d, "**", "*kernel_trace.csv"))
        if not f:
            raise FileNotFoundError(f"no csv under {d}")
        steps = own["steps"] if own["roofline"]["launches"] == 1 else 1
        return read_rows(f), own["roofline"]["kernel_ms"] * 1e3, steps

    def do_stats():
        d = os.path.join(G, f"{tag}_bench_stats")
        st = newest(os.path.join(d, "**", "*kernel_stats.csv"))
        if not st:
            raise FileNotFoundError(f"no kernel_stats.csv under {d}")
        rows, expect_us, steps = bench_pass("bench_stats")
        tr = newest(os.path.join(d, "**", "*kernel_trace.csv"))
        good = timed_dispatches(read_rows(tr), base, dtype, expect_us, "bench_stats")
        shutil.copy(st, os.path.join(P, f"{tag}_bench_kernel_stats.csv"))
        log(f"kernel stats: {len(good)} timed dispatches of {base}{dtype_token(dtype)}..>, mean "
            f"{sum(g['ns'] for g in good.values()) / len(good) * 1e-3:.1f} us (events: {expect_us:.1f})")

    def do_hbm():
        rec = {"source": "rocprofv3 --kernel-trace --pmc <X> --output-format csv -- python3 bench.py --no-cpu --no-extra (separate "
                         "passes for FETCH_SIZE and WRITE_SIZE), tools/collect.sh; dispatches selected by instantiation and "
                         "duration (tools/profile_summary.py)",
               "kernel": bench["roofline"]["kernel"], "workload": workload}
        per = {}
        for cname, pname in (("FETCH_SIZE", "bench_fetch"), ("WRITE_SIZE", "bench_write")):
            rows, expect_us, steps = bench_pass(pname)
            good = timed_dispatches(rows, base, dtype, expect_us, pname)
            per[cname] = mean_counters(good)[cname]
            rec["steps_per_launch"] = steps
            rec[f"{cname}_dispatches"] = len(good)
            rec[f"{cname}_dispatch_us"] = round(sum(g["ns"] for g in good.values()) / len(good) * 1e-3, 1)
            rec[f"{cname}_kernel_name"] = next(iter(good.values()))["name"][:80]
        fk, wk = per["FETCH_SIZE"], per["WRITE_SIZE"]
        rec.update({"sim_path": 2 if "persistent" in bench["roofline"]["kernel"] else 1,
                    "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                    "hbm_bytes_per_launch_corrected": int(2 * fk * 1024 + wk * 1024),
                    "hbm_bytes_per_rod_step": round((2 * fk * 1024 + wk * 1024) / (B * rec["steps_per_launch"]), 1),
                    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE on gfx950 counts 1/2 of the bytes of 16-B-per-lane "
                                  "reads -> doubled (upper bound here: the kernel only reads its initial state and the controls); "
                                  "WRITE_SIZE exact for 16-B-per-lane stores; counters are in KB"})
        json.dump(rec, open(os.path.join(P, f"{tag}_pmc_hbm.json"), "w"), indent=1)
        log(f"HBM: {rec['hbm_bytes_per_rod_step']} B per rod-step ({rec['hbm_bytes_per_launch_corrected']} per launch)")

    def do_sq():
        rows, expect_us, steps = bench_pass("bench_sq")
        good = timed_dispatches(rows, base, dtype, expect_us, "bench_sq")
        sq = mean_counters(good)
        rec = {"source": "rocprofv3 --kernel-trace --pmc SQ_* (one pass), tools/collect.sh", "kernel": bench["roofline"]["kernel"],
               "workload": workload, "steps_per_launch": steps, "per_launch": sq, "launches_profiled": len(good),
               "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)"}
        if sq.get("SQ_WAVE_CYCLES"):
            rec["valu_active_fraction_of_wave_cycles"] = sq.get("SQ_ACTIVE_INST_VALU", 0) / sq["SQ_WAVE_CYCLES"]
        json.dump(rec, open(os.path.join(P, f"{tag}_pmc_sq.json"), "w"), indent=1)
        log("SQ: " + str({k: f"{v:.3g}" for k, v in sq.items()}))

    ops = {"source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 / _MFMA_MOPS_* (tools/collect.sh, sections "
                     "ops / cfg2 / cfg5 / nn); vector flops = 64 lanes x (ADD + MUL + 2 FMA + TRANS) wave-instructions (upper bound: "
                     "not every lane is active), matrix flops = 512 x MOPS",
           "legs": {}}

    def ops_entry(name, c, units, us, dt, waves_note=None):
        fl = executed_flops(c)
        tot_v = fl["valu_f64"] + fl["valu_f32"]
        e = {"units_per_launch": units, "dispatch_us": round(us, 1), "counters_per_launch": c,
             "executed_flops_per_unit": {k: round(v / units, 1) for k, v in fl.items() if v},
             "valu_insts_per_unit": round(c.get("SQ_INSTS_VALU", 0) / units, 1)}
        t = us * 1e-6
        if dt == "f64":
            e["valu_fp64_tflops"] = round(fl["valu_f64"] / t / 1e12, 2)
            e["valu_fp64_frac_of_peak"] = round(fl["valu_f64"] / t / 1e12 / PEAK_TF["F64"], 4)
        else:
            e["valu_fp32_tflops"] = round(fl["valu_f32"] / t / 1e12, 2)
            e["valu_fp32_frac_of_peak"] = round(fl["valu_f32"] / t / 1e12 / PEAK_TF["F32"], 4)
        if fl["mfma_f64"] or fl["mfma_f32"] or fl["mfma_bf16"]:
            # time the matrix pipes would need at their dense peaks for what was executed, over the launch time
            e["mfma_tflops"] = {k: round(fl[k] / t / 1e12, 2) for k in ("mfma_f64", "mfma_f32", "mfma_bf16") if fl[k]}
            e["mfma_frac_of_peak"] = round(sum(fl[f"mfma_{k.lower()}"] / PEAK_TF[k] for k in PEAK_TF) / t / 1e12, 4)
        if waves_note:
            e["note"] = waves_note
        ops["legs"][name] = e
        log(f"ops {name}: {e['executed_flops_per_unit']} per unit, {e.get('valu_fp64_frac_of_peak', e.get('valu_fp32_frac_of_peak'))} of the vector peak"
            + (f", matrix {e['mfma_frac_of_peak']}" if "mfma_frac_of_peak" in e else ""))

    def do_ops():
        rows, expect_us, steps = bench_pass("bench_ops")
        good = timed_dispatches(rows, base, dtype, expect_us, "bench_ops")
        ops_entry("headline", mean_counters(good), B * steps, sum(g["ns"] for g in good.values()) / len(good) * 1e-3, dtype)

    def leg_pass(leg, pname):
        """(rows, leg record, expected us per timed dispatch) of one pass over tools/leg_only.py <leg>."""
        d = os.path.join(G, f"{tag}_{pname}")
        own = last_json_line(d + ".out", "legjson")
        if own is None:
            raise FileNotFoundError(f"{d}.out holds no legjson line")
        f = newest(os.path.join(d, "**", "*counter_collection.csv")) or newest(os.path.join(d, "**", "*kernel_trace.csv"))
        if not f:
            raise FileNotFoundError(f"no csv under {d}")
        return read_rows(f), own, own["kernel_ms_per_step"] * 1e3 * own["T"]

    kernels = {}

    def do_leg(leg, sim_kernel_names):
        rec = {"what": f"tools/leg_only.py {leg} = bench.py extra.{leg} (same function, same arguments)"}
        d = os.path.join(G, f"{tag}_{leg}_stats")
        own = last_json_line(d + ".out", "legjson")
        if own is None:
            raise FileNotFoundError(f"{d}.out holds no legjson line")
        kbase = kernel_base(own["kernel"])
        dt = own["dtype"]
        rec["leg"] = {k: own[k] for k in ("ms_per_step", "kernel_ms_per_step", "B", "N", "T", "dtype", "kernel", "unconverged", "repeats")}
        expect_us = own["kernel_ms_per_step"] * 1e3 * own["T"]
        tr = newest(os.path.join(d, "**", "*kernel_trace.csv"))
        if tr:
            # every dispatch of the timed call's length: the spread over the repeats is what a single-repeat profile of earlier
            # rounds reported as a second "cfg5 number" (the first timed call of a process runs at a lower shader clock)
            typed = group_dispatches(read_rows(tr), kbase, dtype_token(dt))
            T = own["T"]
            # (a leg's untimed warm-up call is a dispatch of the same kernel over fewer steps: only launches of the timed length)
            timed = sorted((k, d_["ns"] * 1e-3) for k, d_ in typed.items() if abs(d_["ns"] * 1e-3 - expect_us) <= 0.35 * expect_us)
            rec["timed_dispatches_us_per_step"] = [round(us / T, 2) for _, us in timed]
            rec["events_best_us_per_step"] = round(own["kernel_ms_per_step"] * 1e3, 2)
            if timed:
                best = min(us for _, us in timed)
                if abs(best - expect_us) > TOL * expect_us:
                    raise ProfileMismatch(f"{leg}_stats: best timed dispatch {best:.1f} us against {expect_us:.1f} us by HIP events")
        for cname, pname in (("FETCH_SIZE", f"{leg}_fetch"), ("WRITE_SIZE", f"{leg}_write")):
            try:
                rows, o2, e_us = leg_pass(leg, pname)
            except FileNotFoundError:
                continue
            good = timed_dispatches(rows, kernel_base(o2["kernel"]), o2["dtype"], e_us, pname)
            rec.setdefault("hbm", {})[cname + "_KB_per_launch"] = mean_counters(good)[cname]
        if "hbm" in rec and len(rec["hbm"]) == 2:
            f_, w_ = rec["hbm"]["FETCH_SIZE_KB_per_launch"], rec["hbm"]["WRITE_SIZE_KB_per_launch"]
            tot = 2 * f_ * 1024 + w_ * 1024
            units = own["B"] * own["T"]
            es = 8 if dt == "f64" else 4
            rec["hbm"].update({"hbm_bytes_per_launch_corrected": int(tot), "hbm_bytes_per_rod_step": round(tot / units, 1),
                               "algorithmic_bytes_per_rod_step": {"tip_only_ring": 7 * es, "full_trajectory": (25 * own["N"] + 4) * es},
                               "GBs_while_running": round(tot / (expect_us * 1e-6) / 1e9, 1)})
        merged_ops, merged_meta = {}, None
        for pname, key in ((f"{leg}_sq", "sq"), (f"{leg}_ops", "ops"), (f"{leg}_vops", "ops")):
            try:
                rows, o2, e_us = leg_pass(leg, pname)
            except FileNotFoundError:
                continue
            good = timed_dispatches(rows, kernel_base(o2["kernel"]), o2["dtype"], e_us, pname)
            c = mean_counters(good)
            us = sum(g["ns"] for g in good.values()) / len(good) * 1e-3
            if key == "sq":
                wc = c.get("SQ_WAVE_CYCLES")
                r2 = {"per_launch": c, "dispatches": len(good)}
                if wc:
                    r2["per_wave_cycle"] = {"mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc), 4),
                                            "valu_active": round(c.get("SQ_ACTIVE_INST_VALU", 0) / wc, 4),
                                            "wait_any": round(c.get("SQ_WAIT_ANY", 0) / wc, 4)}
                rec["sq"] = r2
            else:  # (matrix and vector operation counters come from two passes over the same leg: one entry)
                merged_ops.update(c)
                merged_meta = (o2["B"] * o2["T"], us, o2["dtype"])
        if merged_meta:
            ops_entry(leg, merged_ops, *merged_meta)
        kernels[leg] = rec

    def do_train(cfg):
        d = os.path.join(G, f"{tag}_train_{cfg}_stats")
        st = newest(os.path.join(d, "**", "*kernel_stats.csv"))
        if not st:
            raise FileNotFoundError(f"no kernel_stats.csv under {d}")
        keep = ("mlp_", "loss", "adam", "pack_", "reduce_", "tail")
        rec = {"what": f"tools/train_only.py {cfg}: 20 training epochs through kr_train_epoch"}
        rec["kernel_stats"] = [{"kernel": r["Name"][:110], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                                "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3)}
                               for r in read_rows(st) if any(k in r["Name"] for k in keep)]
        lines = [f"== {cfg}"] + [f"{k['kernel'][:60]:60s} calls {k['calls']:5d} avg {k['avg_us']:8.1f} us" for k in rec["kernel_stats"]]
        avg = {k["kernel"].split("<")[0].split("::")[-1].split("(")[0]: k["avg_us"] for k in rec["kernel_stats"]}
        for pname in ("sq1", "sq2", "fetch", "write"):
            f = newest(os.path.join(G, f"{tag}_train_{cfg}_{pname}", "**", "*counter_collection.csv"))
            if not f:
                continue
            rows = read_rows(f)
            for kname, us in avg.items():
                if not kname.startswith("mlp_") and "tail" not in kname:
                    continue
                # training kernels are all fp32: the duration check alone separates them from any other program's launches
                typed = group_dispatches(rows, kname + "<", "") or group_dispatches(rows, kname, "")
                good = {k: d_ for k, d_ in typed.items() if abs(d_["ns"] * 1e-3 - us) <= 0.3 * us}
                if not good:
                    continue
                c = mean_counters(good)
                rec.setdefault("counters", {}).setdefault(kname, {}).update({k: round(v, 1) for k, v in c.items()})
        for kname, c in rec.get("counters", {}).items():
            wc = c.get("SQ_WAVE_CYCLES")
            if wc:
                c["mfma_busy_per_wave_cycle"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * wc), 4)
            lines.append(f"{kname}: " + json.dumps(c))
        kernels[f"train_{cfg}"] = rec
        return lines

    section(do_stats)
    section(do_hbm)
    section(do_sq)
    section(do_ops)
    for leg in ("cfg2", "cfg5", "cfg3_nn_f64", "cfg3_nn_f32", "headline_full_trajectory"):
        section(lambda leg=leg: do_leg(leg, None))
    # (the nn legs are collected under the pass names nn_f64_* / nn_f32_*: same code, mapped here)
    train_lines = []
    for cfg in ("cfg3", "cfg4", "literal"):
        section(lambda cfg=cfg: train_lines.extend(do_train(cfg)))
    st = newest(os.path.join(G, f"{tag}_ode_stats", "**", "*kernel_stats.csv"))
    if st:
        rec = {"what": "tools/aux_bench.py ode: kr_ode_batch, 78 sizeof(T) algorithmic bytes per row",
               "kernel_stats": [{"kernel": r["Name"][:110], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2)}
                                for r in read_rows(st) if "ode_batch" in r["Name"]]}
        t = os.path.join(G, f"{tag}_ode.txt")
        if os.path.exists(t):
            rec["lines"] = [l.strip() for l in open(t) if l.startswith("ode_batch")]
        kernels["ode_batch"] = rec
    if kernels:
        json.dump(kernels, open(os.path.join(P, f"{tag}_kernels.json"), "w"), indent=1)
    if ops["legs"]:
        json.dump(ops, open(os.path.join(P, f"{tag}_pmc_ops.json"), "w"), indent=1)
    if train_lines:
        open(os.path.join(P, f"{tag}_train.txt"), "w").write("\n".join(train_lines) + "\n")
    src = os.path.join(G, f"{tag}_configs.txt")
    if os.path.exists(src):
        open(os.path.join(P, f"{tag}_configs.txt"), "w").write("".join(l for l in open(src) if "amdgpu.ids" not in l))
    log("profiles/ now holds: " + str(sorted(f for f in os.listdir(P) if f.startswith(tag))))
    return problems


if __name__ == "__main__":
    tag = sys.argv[1]
    bad = summarise(tag, os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles"))
    if bad:
        print(f"{len(bad)} section(s) refused", file=sys.stderr)
        sys.exit(1)
