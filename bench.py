#!/usr/bin/env python3
"""Headline benchmark: rod-steps/s of the forward Cosserat-rod simulation.

Workload (BASELINE.json metric "rod-steps/sec (N=100 segments, batch=1024)"):
per GPU B=1024 rods, N=100 grid points, explicit-Euler shooting sweep inside an
implicit BDF2 time step, fp64 (the reference's NumPy path is fp64), NN off,
setup_robot(mod=None) parameters, per-rod sinusoidal tendon tensions
(SURVEY 8d cfg3 inputs, default_rng(1235)).  One bench "step" = one time step of
the whole batch; the K timed steps are one kr_simulate_batch call (state resident in
HBM as a 3-slot ring of packed states; at this batch size the library runs them
as ONE persistent launch in which every wavefront keeps its rod).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--nodes-per-rod N] [--dtype f64|f32]

The batch is ONE trajectory per rod from the straight rod: `settle` + W untimed steps, then the K timed
ones.  W is what the command line asks for; `settle` (reported separately) tops the untimed part up to 30
steps, the time the start-value predictor of the solver needs to reach its steady state.  What a cold
start costs is reported next to it (`cold_start`: T = 64 and T = 200 from the straight rod, no hand-over).
Rods 0..31 of the timed batch are compared with the CPU oracle over the untimed AND the first timed steps.

N>1: launched by torch.distributed.run, one rank per GPU; rods are sharded
(weak scaling, no data-path collective); the only collectives are the timing
barrier and the MAX over ranks of the elapsed time.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_VALU_PEAK_TF = 78.6   # public MI355X fp64 vector peak = 1024 SIMDs x 16 FMA lanes x 2 flop x 2.4 GHz
SIMDS, CLOCK_HZ = 1024, 2.4e9
SETTLE_TOTAL = 30          # untimed steps the predictor needs
SEED = 1235


def rank_controls(B, world, rank, steps, del_t, first_step=1):
    """Tensions [B, steps, 4] of this rank's rods: rod b of the GLOBAL batch of world x B rods has
    period P_b ~ U[0.5, 3] s and phase phi_b ~ U[0, 2 pi) from default_rng(1235); rank r owns rods
    [r B, (r + 1) B).  Step i (1-based from the straight rod) applies 6 + sin(2 pi i dt / P_b + phi_b + k pi / 2)."""
    import numpy as np
    rng = np.random.default_rng(SEED)
    Pd = rng.uniform(0.5, 3.0, size=B * world)[rank * B:(rank + 1) * B]
    phi = rng.uniform(0.0, 2 * np.pi, size=B * world)[rank * B:(rank + 1) * B]
    k = np.arange(4)[None, None, :]
    i = np.arange(first_step, first_step + steps)[None, :, None]
    return 6.0 + np.sin(2 * np.pi * i * del_t / Pd[:, None, None] + phi[:, None, None] + k * (np.pi / 2))


def committed_profile(B, N, dtype, path, kind):
    """Per-rod-step figures from the newest committed rocprofv3 --pmc summary of this workload
    (profiles/*pmc_<kind>.json, written by tools/summarise_profiles.py); None if there is none."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_{kind}.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload", f"B={B} N={N} {dtype} Euler") != f"B={B} N={N} {dtype} Euler":
            continue
        if kind == "hbm" and d.get("sim_path", 2) != path:
            continue
        d["_file"] = os.path.basename(f)
        best = d
    return best


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(N, del_t, B, sample_steps, rods=None):
    """Times the oracle (NumPy port of cosserat_ode.py + knode.simulate with scipy fsolve - the reference's own
    execution model) on the host: one rod per process on every available core, rods 0.. of rank 0's TIMED batch,
    from the straight rod.  Median of 3 runs.  Returns (record, tips[rods][steps][3])."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import multiprocessing as mp
    cores = min(len(os.sched_getaffinity(0)), 16)
    rods = rods or 2 * cores  # >= 32 rods x 50 steps on all host cores
    ctl = rank_controls(B, 1, 0, sample_steps, del_t)[:rods]
    t0 = time.perf_counter()
    _cpu_worker((N, ctl[0]))
    t_single = time.perf_counter() - t0
    rates, tips = [], None
    with mp.get_context("fork").Pool(cores) as pool:
        for _ in range(3):
            t0 = time.perf_counter()
            tips = pool.map(_cpu_worker, [(N, ctl[b]) for b in range(rods)], chunksize=1)
            rates.append(rods * sample_steps / (time.perf_counter() - t0))
    return {
        "value": round(statistics.median(rates), 3),
        "unit": "rod-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"rods 0..{rods - 1} of the timed batch x {sample_steps} steps from the straight rod (N={N}, fp64, "
                  f"fsolve shooting), one rod per process, {cores} processes, median of 3 runs",
        "runs": [round(r, 3) for r in rates],
        "single_core_value": round(sample_steps / t_single, 3),
        "cpu_model": cpu_model(),
    }, tips


def _cpu_worker(args):
    N, ctl = args
    os.environ["OMP_NUM_THREADS"] = "1"
    import warnings
    import numpy as np
    import cosserat_oracle as orc
    D = orc.params_for(None, N).derived()
    # the oracle mirrors knode.simulate: T controls -> T solves, last one dropped from the output.  Its fsolve probes
    # overflow on the way (the reference's does too): keep that off stderr, the JSON line must be the last thing printed
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        traj = orc.simulate(D, np.vstack([ctl, ctl[-1:]]), solver="fsolve")
    return traj[1:, :3, -1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--batch", type=int, default=1024, help="rods per GPU")
    ap.add_argument("--nodes-per-rod", type=int, default=100, help="N, grid points per rod")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg and the cold-start runs")
    args = ap.parse_args()

    import numpy as np
    import torch
    import krod_native as kn  # noqa: F401
    from cosserat_ode import CosseratRod
    from knode import setup_robot

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")

    # ---- CPU leg FIRST: it forks a process pool, and a process that has initialised the GPU must not be forked (nor
    # ever exec'd).  Nothing above touches HIP: importing torch / loading libknode_rod.so does not, and the parameter
    # presets are host-side library calls.
    cpu_leg = None
    if world == 1 and not args.no_cpu:
        probe = CosseratRod(use_fsolve=True)
        setup_robot(probe)
        pre0 = max(0, SETTLE_TOTAL - args.warmup) + args.warmup
        Tc = min(pre0 + args.steps, max(50, pre0 + 20))
        cb, cpu_tips = cpu_baseline(args.nodes_per_rod, probe.del_t, args.batch, Tc)
        try:
            cbc = cpu_baseline_c(args.nodes_per_rod, probe.del_t, args.batch)
        except Exception as e:  # the C restatement is optional test infrastructure (needs gcc or its prebuilt .so)
            cbc = {"error": str(e)}
        cpu_leg = (Tc, cb, cpu_tips, cbc)
    # one rank per GPU; KR_BENCH_BACKEND=gloo lets several ranks share one GPU for a rehearsal of the
    # multi-rank code path on a 1-GPU box (collectives then run on CPU tensors)
    backend = os.environ.get("KR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = f"cuda:{dev_index}"
    cdev = dev if backend == "nccl" else "cpu"  # where the collective's tensors live
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))  # before any other GPU call of this process
        else:
            dist.init_process_group(backend)

    B, N, K, W = args.batch, args.nodes_per_rod, args.steps, args.warmup
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    esize = 8 if args.dtype == "f64" else 4

    robot = CosseratRod(use_fsolve=True, device=dev_index)
    setup_robot(robot)
    robot.N = N
    robot.compute_intermediate_terms()
    h = robot._native()

    settle = max(0, SETTLE_TOTAL - W)
    pre = settle + W  # untimed steps of the trajectory
    ctl = rank_controls(B, world, rank, pre + K, robot.del_t)
    ctl_pre = torch.as_tensor(ctl[:, :pre], device=dev).to(tdt).contiguous() if pre else None
    ctl_k = torch.as_tensor(ctl[:, pre:], device=dev).to(tdt).contiguous()
    G = torch.zeros((B, 6), dtype=tdt, device=dev)
    status = torch.zeros((B, K), dtype=torch.int32, device=dev)
    tip = torch.empty((B, K, 3), dtype=tdt, device=dev)
    tip_pre = torch.empty((B, max(pre, 1), 3), dtype=tdt, device=dev)

    # a second handle (own predictor image, own options) for everything that is not the measured trajectory
    robot2 = CosseratRod(use_fsolve=True, device=dev_index)
    setup_robot(robot2)
    robot2.N = N
    robot2.compute_intermediate_terms()
    h2 = robot2._native()

    def run_cold(T, dt=None, scheme=0):
        """T steps from the straight rod with no predictor hand-over: (seconds, unconverged rod-steps)."""
        dt = dt or tdt
        c = torch.as_tensor(rank_controls(B, world, rank, T, robot.del_t), device=dev).to(dt).contiguous()
        st = h2.new_state(B, dt, n_slots=3)
        g0 = torch.zeros((B, 6), dtype=dt, device=dev)
        stat = torch.zeros((B, T), dtype=torch.int32, device=dev)
        h2.init_straight(st[0])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h2.simulate(c, st, g0, ring=True, status=stat, scheme=scheme)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3, int((stat != 0).sum())

    # Clock ramp, not part of the untimed steps.  The shader clock of an MI355X under this fp64 load settles over
    # SECONDS of accumulated load, not milliseconds (tools/clock_probe*.py: the same 20-step launch, every rod at
    # exactly 2 sweeps per step and 84 k ticks per step each time, takes 46 us per step in a process that has just
    # started and 36 us after ~2 s of fp64 work; an fp32 ramp warms less).  The ramp runs the persistent solver with
    # RK4 sweeps in the precision of the measurement on scratch copies of the problem: the same instruction mix, but a
    # different kernel instantiation, so that in a rocprofv3 trace of `bench.py --no-cpu` the timed kernel appears
    # exactly once (the K timed steps); the untimed steps of the trajectory go through the one-launch-per-step form
    # for the same reason.
    h.set_option("keep_predictor", 0)
    persistent_default = h.get_option("persistent")

    ramp_ctl = torch.as_tensor(rank_controls(B, world, rank, 100, robot.del_t), device=dev).to(tdt).contiguous()
    ramp_st = h2.new_state(B, tdt, n_slots=3)
    ramp_g = torch.zeros((B, 6), dtype=tdt, device=dev)

    def ramp(seconds):
        """Queues launches without waiting in between, so that the GPU goes from the last one straight into whatever
        follows the next synchronize()."""
        n = max(1, int(seconds / 0.012))  # one 100-step RK4 launch of this batch takes ~12 ms
        for _ in range(n):
            h2.init_straight(ramp_st[0])
            ramp_g.zero_()
            h2.simulate(ramp_ctl, ramp_st, ramp_g, ring=True, scheme=1)  # KR_RK4
        torch.cuda.synchronize()

    RAMP_A, RAMP_B = 0.5, 0.3
    ramp(RAMP_A)
    # First use of a kernel instantiation and of the handle's per-batch scratch costs host time (symbol lookup in a
    # 10 MB code object, one hipMalloc: ~0.4 ms) that must not sit between the timing events: kr_simulate_prepare does
    # that work ahead of time without launching anything, so the timed kernel still appears once in a trace.
    h.simulate_prepare(B, tdt)
    torch.cuda.synchronize()
    cold = None
    if not args.no_cpu:
        cold = {}
        for T in (64, 200):
            secs, bad = min(run_cold(T) for _ in range(3))
            cold[f"T{T}"] = {"value": round(B * T / secs, 1), "ms_per_step": round(secs / T * 1e3, 4), "unconverged": bad}

    # the untimed and the timed steps are one trajectory advanced by two calls: the second call resumes the
    # start-value predictor of the first (option "keep_predictor").  What precedes the timed launch sets its clock
    # (tools/clock_probe3.py, 20-step launch: 35 us per step right after light launches, 36-37 after 1 s of full load,
    # 40 after 3 s of it, 42 after 20 ms of idling): a short ramp, then the untimed steps (one launch per step: light),
    # then the timed launch with nothing but the contract's barrier + synchronize in between.
    h.set_option("persistent", 0)
    h.set_option("keep_predictor", 1)
    pre_states = h.new_state(B, tdt, n_slots=pre + 1) if pre else None
    states = h.new_state(B, tdt, n_slots=3)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ramp(RAMP_B)  # (on the second handle: the predictor image of `h` is not touched)
    if pre:
        h.init_straight(pre_states[0])
        h.simulate(ctl_pre, pre_states, G, tip=tip_pre)
        states[0].copy_(pre_states[pre])
        prev_init = pre_states[pre - 1]
    else:
        h.init_straight(states[0])
        prev_init = None
    h.set_option("persistent", persistent_default)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    ev0.record()
    h.simulate(ctl_k, states, G, ring=True, tip=tip, status=status, prev_init=prev_init)
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_ms = ev0.elapsed_time(ev1)  # duration of the K-step region on the launch stream (HIP events)
    h.set_option("keep_predictor", 0)

    n_bad = int((status != 0).sum())

    if rank == 0:
        rod_steps = world * B * K
        path = h.get_option("last_sim_path")  # 2: the K steps ran as one persistent launch
        persistent = path == 2
        wpr = h.get_option("last_waves_per_rod")
        if wpr > 1:  # small batches: several wavefronts per rod (kr_msw_impl.hpp)
            kernel_name = (f"kr::msw_sim_kernel (persistent, {wpr} wavefronts per rod)" if persistent
                           else f"kr::msw_step_kernel ({wpr} wavefronts per rod)")
        elif path == 2 and h.get_option("last_overlap"):
            # kr_mso_impl.hpp: one sweep per step in the steady state; the plain persistent kernel is launched behind it
            # for rods that left steps behind (none here: unconverged_rod_steps) and exits at once otherwise
            kernel_name = "kr::mso_sim_kernel (persistent, overlapped steps, all K steps in one launch)"
        else:
            kernel_name = ("kr::step_kernel", "kr::ms_step_kernel",
                           "kr::ms_sim_kernel (persistent, all K steps in one launch)")[path] if path in (0, 1, 2) else "?"
        # SURVEY 8d algorithmic bytes per rod-step: state written every step (25 N + 4) s in the persistent
        # form (history never leaves the CU); (75 N + 16) s when every step is its own launch
        per_rod_step = (25 * N + 4) * esize if persistent else (75 * N + 16) * esize
        launches = 1 if persistent else K
        units_per_launch = B * (K if persistent else 1)  # rod-steps one launch processes
        alg_bytes = units_per_launch * per_rod_step
        kernel_ms = kernel_ms / launches
        hbm_achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # SURVEY 8d algorithmic flops per rod-step: (N-1) (7k+1) F_ode with k = 3 Newton iterations, F_ode = 450
        flops_per_rod_step = (N - 1) * 22 * 450
        tf_achieved = units_per_launch * flops_per_rod_step / (kernel_ms * 1e-3) / 1e12
        prof_hbm = committed_profile(B, N, args.dtype, path, "hbm")
        prof_sq = committed_profile(B, N, args.dtype, path, "sq")
        traffic = None
        if prof_hbm:
            per_unit = prof_hbm["hbm_bytes_per_launch_corrected"] / (B * prof_hbm.get("steps_per_launch", 1))
            traffic = int(per_unit * units_per_launch)
        valu_issue = None
        instr_per_unit = None
        if prof_sq and prof_sq.get("per_launch", {}).get("SQ_INSTS_VALU") and prof_sq.get("steps_per_launch"):
            instr_per_unit = prof_sq["per_launch"]["SQ_INSTS_VALU"] / (B * prof_sq["steps_per_launch"])
            # wave-instructions x 4 issue cycles over the SIMD-cycles of the launch (B <= 1024: one wavefront per SIMD)
            valu_issue = round(instr_per_unit * units_per_launch * 4 / (min(B, SIMDS) * kernel_ms * 1e-3 * CLOCK_HZ), 4)
        out = {
            "metric": "rod-steps/sec (N=100 segments, batch=1024)",
            "value": round(rod_steps / elapsed, 1),
            "unit": "rod-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "settle_steps": settle,
            "ramp_s": RAMP_A + RAMP_B,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"forward simulate, B={B} rods/GPU, N={N}, Euler shooting + BDF2, NN off, "
                            f"setup_robot(None), per-rod sine tensions rng({SEED})",
                "rods_per_gpu": B, "N": N, "unconverged_rod_steps": n_bad,
            },
            "roofline": {
                "bound": "valu_fp64" if args.dtype == "f64" else "valu_issue",
                "achieved": round(tf_achieved, 3),
                "peak": FP64_VALU_PEAK_TF,
                "unit": "TFLOP/s",
                "frac": round(tf_achieved / FP64_VALU_PEAK_TF, 5),
                "traffic": traffic,
                "traffic_scaled_from_profile": traffic,
                "traffic_note": "PMC HBM bytes per rod-step of the committed profile (profile.hbm) x the rod-steps of "
                                "this launch - not a counter pass of this run",
                "algorithmic_flops_per_rod_step": flops_per_rod_step,
                "achieved_is": "nominal: SURVEY 8d formula flops (k = 3 FD-Newton iterations), not executed flops",
                "executed_valu_insts_per_rod_step": instr_per_unit and round(instr_per_unit, 1),
                "valu_issue_frac": valu_issue,
                "hbm": {"achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(hbm_achieved / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": alg_bytes,
                        "algorithmic_bytes_per_rod_step": per_rod_step},
                "kernel": kernel_name,
                "kernel_ms": round(kernel_ms, 4),
                "launches": launches,
                "profile": {"hbm": prof_hbm and prof_hbm["_file"], "sq": prof_sq and prof_sq["_file"]},
                "note": "one rod per wavefront on each of the 1024 SIMDs: the launch is bound by the fp64 vector "
                        "issue rate of a single wave, not by HBM (counter traffic ~ the algorithmic bytes); `achieved` "
                        "prices SURVEY 8d's algorithmic flops per rod-step against the fp64 vector peak, "
                        "`valu_issue_frac` is measured VALU issue (SQ_INSTS_VALU x 4 cycles over SIMD-cycles, "
                        "profiles/*pmc_sq.json), `hbm` the nominal roof of SURVEY 8d",
            },
        }
        if cold is not None:
            out["cold_start"] = {"unit": "rod-steps/s", **cold,
                                 "note": "T steps from the straight rod in one call, no warm-up, no predictor hand-over "
                                         "(SURVEY 8d cfg3: T=64, cfg2: T=200), best of 3"}
        if cpu_leg is not None:
            Tc, cb, tips, cbc = cpu_leg
            # tip parity of the TIMED batch against the oracle: same rods, same steps (untimed + first timed ones)
            gpu_tips = torch.cat([tip_pre[:, :pre], tip], dim=1)[: len(tips), :Tc].double().cpu().numpy()
            errs = [float(np.linalg.norm(gpu_tips[b] - tips[b]) / np.linalg.norm(tips[b])) for b in range(len(tips))]
            out["tip_rel_l2_vs_oracle"] = max(errs)
            out["tip_check"] = {"rods": len(tips), "steps": Tc, "timed_steps_included": max(0, Tc - pre),
                                "what": "rods 0.. of the timed batch, steps 1..steps of their trajectory"}
            out["cpu_baseline"] = cb
            out["cpu_baseline_c"] = cbc
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline_c(N, del_t, B, rods=256, steps=150):
    """The scalar C restatement (oracle/cosserat_oracle_c.c, Newton shooting) on all host cores, one rod per call,
    threads (the C call releases the GIL): what an optimised CPU implementation of the same discrete equations
    reaches, beside the NumPy port that has the reference's own execution model."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from concurrent.futures import ThreadPoolExecutor
    import cosserat_oracle as orc
    import cosserat_oracle_c as oc
    cores = min(len(os.sched_getaffinity(0)), 16)
    P = orc.params_for(None, N)
    ctl = rank_controls(max(B, rods), 1, 0, steps, del_t)[:rods]
    oc.simulate(P, ctl[0][:4], traj=False)  # load + build outside the timed region
    t0 = time.perf_counter()
    one = oc.simulate(P, ctl[0], traj=False)
    t_single = time.perf_counter() - t0
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        res = list(ex.map(lambda c: oc.simulate(P, c, traj=False), ctl))
        t_all = time.perf_counter() - t0
    return {"value": round(rods * steps / t_all, 1), "unit": "rod-steps/s", "cores": cores, "kind": "port",
            "sample": f"{rods} rods x {steps} steps of the bench workload (N={N}, fp64), scalar C, Newton shooting to 1e-12, "
                      f"one rod per thread", "single_core_value": round(steps / t_single, 1),
            "unconverged": int(sum(r[2] for r in res) + one[2])}


if __name__ == "__main__":
    main()
