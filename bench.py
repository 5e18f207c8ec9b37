#!/usr/bin/env python3
"""Headline benchmark: rod-steps/s of the forward Cosserat-rod simulation, plus one sub-object per other
BASELINE.json configuration (``extra``) and - with ``--gpus N > 1`` - the data-parallel training leg with the RCCL
gradient all-reduce (``extra.train_dp``).

Headline workload (BASELINE.json metric "rod-steps/sec (N=100 segments, batch=1024)"): per GPU B=1024 rods, N=100 grid
points, explicit-Euler shooting sweep inside an implicit BDF2 time step, fp64 (the reference's NumPy path is fp64), NN
off, setup_robot(mod=None) parameters, per-rod sinusoidal tendon tensions (SURVEY 8d cfg3 inputs, default_rng(1235)).
One bench "step" = one time step of the whole batch; K timed steps are one kr_simulate_batch call (state resident in
HBM as a 3-slot ring of packed states; at this batch size the library runs them as ONE persistent launch in which every
wavefront keeps its rod).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--nodes-per-rod N] [--dtype f64|f32]
                    [--scaling weak|strong] [--chunks C] [--no-cpu] [--no-extra]

The batch is ONE trajectory per rod from the straight rod: `settle` + W untimed steps, then C (default 5) consecutive
chunks of K timed steps, each chunk bracketed by barrier + synchronize; `value` is the MEDIAN chunk (min / median / max /
first are reported in `timed_chunks`).  W is what the command line asks for; `settle` (reported separately) tops the
untimed part up to 30 steps, the time the start-value predictor of the solver needs to reach its steady state.  What
a cold start costs is reported next to it (`cold_start`: T = 64 and T = 200 from the straight rod, no hand-over).
Rods 0..31 of the timed batch are compared with the CPU oracle over the untimed AND the first timed steps.

N>1: launched by torch.distributed.run, one rank per GPU; rods are sharded (weak scaling by default: 1024 rods on every
GPU, no data-path collective - the only collectives of the forward leg are the timing barrier and the MAX over ranks
of the elapsed time; `--scaling strong` splits 1024 rods over the GPUs, the mode not chosen is reported under `extra`).
Then BASELINE cfg4 runs as `extra.train_dp`: 4096 trajectories sharded over the ranks, 28->512->25, 50 epochs, ONE
all-reduce (SUM) of the flat fp32 gradient + loss buffer per epoch - RCCL over xGMI with the nccl backend - and the
single-rank repeat of the same global batch on rank 0 as the parity check.
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
CPU_WORKERS_MAX = 16       # CPU share of a one-GPU box of this pool; more worker processes trip its process guard


def rank_controls(B, world, rank, steps, del_t, first_step=1):
    """Tensions [B, steps, 4] of this rank's rods: rod b of the GLOBAL batch of world x B rods has
    period P_b ~ U[0.5, 3] s and phase phi_b ~ U[0, 2 pi) from default_rng(1235); rank r owns rods
    [r B, (r + 1) B).  Step i (1-based from the straight rod) applies 6 + sin(2 pi i dt / P_b + phi_b + k pi / 2)."""
    import bench_legs as bl
    return bl.sine_controls(B, steps, del_t, SEED, world, rank, first_step)


def committed_profile(B, N, dtype, path, kind):
    """Per-rod-step figures from the newest committed rocprofv3 --pmc summary of this workload
    (profiles/*pmc_<kind>.json, written by tools/profile_summary.py, which attributes a counter row to the workload only
    if kernel instantiation AND dispatch duration match the profiled process's own timing); None if there is none."""
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


MFMA_PEAK_TF = {"mfma_f64": 78.6, "mfma_f32": 157.3, "mfma_bf16": 2516.6}


def executed_roofline(leg, units_per_s, dtype):
    """`roofline.executed` of a leg: the arithmetic the kernel actually executed per unit (counter pass of the committed
    profile, profiles/*_pmc_ops.json: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes, SQ_INSTS_VALU_MFMA_MOPS_* x 512) at
    THIS run's rate, against the vector peak of the run's type and - for matrix work - against the dense MFMA peak of each
    operand type (the time the matrix pipes would need at peak over the time the launch took)."""
    import glob
    entry, src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_ops.json"))):
        try:
            e = json.load(open(f)).get("legs", {}).get(leg)
        except Exception:
            continue
        if e:
            entry, src = e, os.path.basename(f)
    if entry is None:
        return None
    fl = entry["executed_flops_per_unit"]
    vkey = "valu_f64" if dtype == "f64" else "valu_f32"
    vpeak = FP64_VALU_PEAK_TF if dtype == "f64" else 157.3
    out = {"source": src, "flops_per_unit": fl, "valu_insts_per_unit": entry.get("valu_insts_per_unit"),
           "how": "counter-measured operations per unit of the committed profile x this run's units per second; vector flops "
                  "assume 64 active lanes per instruction (upper bound: the step kernels run 58 of 64)"}
    if fl.get(vkey):
        tf = fl[vkey] * units_per_s / 1e12
        out["valu"] = {"achieved": round(tf, 3), "peak": vpeak, "unit": "TFLOP/s", "frac": round(tf / vpeak, 5)}
    m = {k: fl[k] * units_per_s / 1e12 for k in MFMA_PEAK_TF if fl.get(k)}
    if m:
        out["mfma"] = {"achieved": {k: round(v, 3) for k, v in m.items()}, "peaks": {k: MFMA_PEAK_TF[k] for k in m}, "unit": "TFLOP/s",
                       "frac": round(sum(v / MFMA_PEAK_TF[k] for k, v in m.items()), 5)}
    out["frac"] = round((out.get("valu", {}).get("frac", 0.0)) + (out.get("mfma", {}).get("frac", 0.0)), 5)
    return out


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
    avail = len(os.sched_getaffinity(0))
    cores = min(avail, CPU_WORKERS_MAX)
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
        "cores_available": avail,
        "cpu_count": os.cpu_count(),
        "cores_note": f"worker processes are capped at {CPU_WORKERS_MAX}: the CPU share of a one-GPU box of this pool "
                      "(the job's process guard); value scales ~linearly with cores for this embarrassingly parallel port",
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


def oracle_refs(del_t, want):
    """Reference tips (rods 0, 1 from the straight rod) for the accuracy field of the extra legs: the scalar C oracle for
    the physics-only configurations, the NumPy oracle's tight Newton solve where the MLP is on (two processes).  CPU
    leg: runs before the first GPU call.  want: {name: (seed, N, steps, B, mlp_sizes or None)}."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import multiprocessing as mp
    import numpy as np
    import bench_legs as bl
    import cosserat_oracle as orc
    import cosserat_oracle_c as oc
    refs, nn_jobs = {}, []
    for name, (seed, N, steps, B, sizes) in want.items():
        ctl = bl.sine_controls(B, steps, del_t, seed)[:2]
        if sizes is None:
            refs[name] = np.stack([oc.simulate(orc.params_for(None, N), ctl[b], traj=False)[0] for b in range(2)])
        else:
            nn_jobs += [(name, N, ctl[b], sizes) for b in range(2)]
    if nn_jobs:
        with mp.get_context("fork").Pool(min(len(nn_jobs), 4)) as pool:
            res = pool.map(_nn_worker, nn_jobs)
        for (name, *_), tips in zip(nn_jobs, res):
            refs.setdefault(name, []).append(tips)
        for name in {j[0] for j in nn_jobs}:
            refs[name] = np.stack(refs[name])
    return refs


def _nn_worker(args):
    name, N, ctl, sizes = args
    os.environ["OMP_NUM_THREADS"] = "1"
    import warnings
    import numpy as np
    import bench_legs as bl
    import cosserat_oracle as orc
    Ws, bs = bl.mlp_weights(sizes, 7)
    mlp = orc.Mlp(Ws, bs, [orc.ACT_ELU] * (len(Ws) - 1) + [orc.ACT_NONE], False)
    D = orc.params_for(None, N).derived()
    with np.errstate(all="ignore"), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        traj = orc.simulate(D, np.vstack([ctl, ctl[-1:]]), mlp=mlp, solver="newton")
    return traj[1:, :3, -1]


NN_SIZES = [28, 64, 64, 25]   # the KNODE network of BASELINE cfg3 in the reference's I/O contract
EXTRA_REFS = {  # name: (seed, N, steps, B, mlp sizes)
    "cfg2": (1234, 100, 12, 256, None),
    "headline_full_trajectory": (SEED, 100, 12, 1024, None),
    "cfg5": (1237, 400, 8, 512, None),
    "cfg3_nn": (SEED, 100, 8, 1024, NN_SIZES),
}


def extra_legs(refs=None):
    """The other BASELINE.json configurations as a table: name -> (function of bench_legs, positional arguments
    after (torch, dev_index), keyword arguments).  bench.py runs every entry under `extra`; tools/leg_only.py runs ONE of
    them under rocprofv3 with the same arguments, so a profile and the driver line describe the same launches."""
    import bench_legs as bl
    refs = refs or {}
    mlp = bl.mlp_weights(NN_SIZES, 7)
    return {
        "cfg2": ("forward_leg", (256, 100, 200, 60, "f64", 1234), {"ref_tips": refs.get("cfg2")}),
        "cfg3_nn_f64": ("forward_leg", (1024, 100, 64, 0, "f64", SEED), {"mlp": mlp, "ref_tips": refs.get("cfg3_nn"), "repeats": 2}),
        "cfg3_nn_f32": ("forward_leg", (1024, 100, 64, 0, "f32", SEED), {"mlp": mlp, "ref_tips": refs.get("cfg3_nn"), "repeats": 2}),
        "cfg3_train_epoch": ("train_leg", (1024, 64, 100, [22, 67, 99], [64, 64]), {}),
        "cfg3_mlp_literal": ("mlp_literal_leg", (193536,), {}),
        "cfg4_shard_epoch": ("train_leg", (512, 30, 10, [3, 5, 7, 9], [512]), {}),
        "cfg5": ("forward_leg", (512, 400, 60, 30, "f64", 1237), {"ref_tips": refs.get("cfg5")}),
        "cfg5_f32": ("forward_leg", (512, 400, 60, 30, "f32", 1237), {"ref_tips": refs.get("cfg5")}),
        "cfg5_tolerance_sweep": ("tolerance_sweep_leg", (), {}),
        "headline_full_trajectory": ("forward_leg", (1024, 100, 200, 60, "f64", SEED),
                                     {"full_trajectory": True, "ref_tips": refs.get("headline_full_trajectory")}),
    }


def run_extras(torch, dev_index, refs, log):
    """Runs every leg of extra_legs() on this GPU, one sub-object each."""
    import bench_legs as bl
    legs = {}
    for name, (fn, a, k) in extra_legs(refs).items():
        t0 = time.perf_counter()
        try:
            legs[name] = getattr(bl, fn)(torch, dev_index, *a, **k)
        except Exception as e:  # a leg that fails must not take the headline line with it
            legs[name] = {"error": f"{type(e).__name__}: {e}"}
        legs[name]["leg_wall_s"] = round(time.perf_counter() - t0, 2)
        r = legs[name].get("roofline")
        if r is not None and fn == "forward_leg" and legs[name].get("kernel_ms_per_step"):
            key = "headline" if name == "headline_full_trajectory" else name
            ex = executed_roofline(key, legs[name]["B"] / (legs[name]["kernel_ms_per_step"] * 1e-3), legs[name]["dtype"])
            r["executed"] = ex
            if r.get("frac") is None and ex is not None:
                # MLP-on legs: the nominal count is not a utilisation (bench_legs.forward_leg); the executed one is
                r["frac"] = ex["frac"]
                r["frac_is"] = "executed (roofline.executed): matrix-pipe time at dense peak per operand type + vector flops at the vector peak, over the launch time"
        log(f"extra {name}: {legs[name].get('value')} {legs[name].get('unit')} ({legs[name]['leg_wall_s']} s)")
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--batch", type=int, default=1024, help="rods per GPU (weak scaling) / rods in all (strong)")
    ap.add_argument("--nodes-per-rod", type=int, default=100, help="N, grid points per rod")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch rods on every GPU; strong: --batch rods split over the GPUs")
    ap.add_argument("--chunks", type=int, default=5, help="consecutive K-step chunks of the trajectory that are timed "
                    "(value = their median)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg and the cold-start runs")
    ap.add_argument("--no-extra", action="store_true", help="skip the legs for the other BASELINE configurations / "
                    "the data-parallel training leg")
    args = ap.parse_args()

    import numpy as np
    import torch
    import krod_native as kn  # noqa: F401
    import bench_legs as bl
    from cosserat_ode import CosseratRod
    from knode import setup_robot

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    log = (lambda m: print("[bench] " + m, file=sys.stderr, flush=True)) if rank == 0 else (lambda m: None)

    # ---- CPU leg FIRST: it forks a process pool, and a process that has initialised the GPU must not be forked (nor
    # ever exec'd).  Nothing above touches HIP: importing torch / loading libknode_rod.so does not, and the parameter
    # presets are host-side library calls.
    cpu_leg = None
    refs = None
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
        if not args.no_extra:
            try:
                refs = oracle_refs(probe.del_t, EXTRA_REFS)
            except Exception as e:
                log(f"oracle references for the extra legs failed: {e}")
        log("CPU leg done")
    # one rank per GPU; KR_BENCH_BACKEND=gloo lets several ranks share one GPU for a rehearsal of the
    # multi-rank code path on a 1-GPU box (collectives then run on CPU tensors)
    backend = os.environ.get("KR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = f"cuda:{dev_index}"
    cdev = dev if backend == "nccl" else "cpu"  # where the collective's tensors live
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))  # before any other GPU call of this process
        else:
            dist.init_process_group(backend)

    N, K, W = args.nodes_per_rod, args.steps, args.warmup
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    esize = 8 if args.dtype == "f64" else 4

    robot = CosseratRod(use_fsolve=True, device=dev_index)
    setup_robot(robot)
    robot.N = N
    robot.compute_intermediate_terms()
    h = robot._native()
    # a second handle (own predictor image, own options) for everything that is not the measured trajectory
    robot2 = CosseratRod(use_fsolve=True, device=dev_index)
    setup_robot(robot2)
    robot2.N = N
    robot2.compute_intermediate_terms()
    h2 = robot2._native()

    settle = max(0, SETTLE_TOTAL - W)
    pre = settle + W  # untimed steps of the trajectory
    n_chunks = max(1, args.chunks)

    def run_cold(B, T, dt=None, scheme=0):
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
    # different kernel instantiation, so that in a rocprofv3 trace of `bench.py --no-cpu --no-extra --chunks 1` the
    # timed kernel appears exactly once (the K timed steps); the untimed steps of the trajectory go through the
    # one-launch-per-step form for the same reason.
    RAMP_A, RAMP_B = 0.5, 0.3

    def ramp(B, seconds, cache={}):
        """Queues launches without waiting in between, so that the GPU goes from the last one straight into whatever
        follows the next synchronize()."""
        if B not in cache:
            cache[B] = (torch.as_tensor(rank_controls(B, world, rank, 100, robot.del_t), device=dev).to(tdt).contiguous(),
                        h2.new_state(B, tdt, n_slots=3), torch.zeros((B, 6), dtype=tdt, device=dev))
        ramp_ctl, ramp_st, ramp_g = cache[B]
        n = max(1, int(seconds / 0.012))  # one 100-step RK4 launch of 1024 rods takes ~12 ms
        for _ in range(n):
            h2.init_straight(ramp_st[0])
            ramp_g.zero_()
            h2.simulate(ramp_ctl, ramp_st, ramp_g, ring=True, scheme=1)  # KR_RK4
        torch.cuda.synchronize()

    def headline(B, b_world, b_rank, with_cold):
        """`pre` untimed steps, then n_chunks x K timed steps of ONE trajectory per rod.  Rods [b_rank B, (b_rank + 1) B)
        of a global draw of b_world x B rods.  Returns the measurement record of this rank (rank 0 assembles)."""
        ctl = rank_controls(B, b_world, b_rank, pre + n_chunks * K, robot.del_t)
        ctl_pre = torch.as_tensor(ctl[:, :pre], device=dev).to(tdt).contiguous() if pre else None
        ctl_k = [torch.as_tensor(ctl[:, pre + c * K: pre + (c + 1) * K], device=dev).to(tdt).contiguous()
                 for c in range(n_chunks)]
        G = torch.zeros((B, 6), dtype=tdt, device=dev)
        status = torch.zeros((n_chunks, B, K), dtype=torch.int32, device=dev)
        tip = torch.empty((n_chunks, B, K, 3), dtype=tdt, device=dev)
        tip_pre = torch.empty((B, max(pre, 1), 3), dtype=tdt, device=dev)
        h.set_option("keep_predictor", 0)
        persistent_default = h.get_option("persistent")
        ramp(B, RAMP_A)
        # First use of a kernel instantiation and of the handle's per-batch scratch costs host time (symbol lookup in a
        # 10 MB code object, one hipMalloc: ~0.4 ms) that must not sit between the timing events: kr_simulate_prepare
        # does that work ahead of time without launching anything, so the timed kernel still appears once in a trace.
        h.simulate_prepare(B, tdt)
        torch.cuda.synchronize()
        cold = None
        if with_cold:
            cold = {}
            for T in (64, 200):
                secs, bad = min(run_cold(B, T) for _ in range(3))
                cold[f"T{T}"] = {"value": round(B * T / secs, 1), "ms_per_step": round(secs / T * 1e3, 4), "unconverged": bad}
        # the untimed and the timed steps are one trajectory advanced by several calls: every call resumes the
        # start-value predictor of the one before (option "keep_predictor").  What precedes a timed launch sets its
        # clock (tools/clock_probe3.py, 20-step launch: 35 us per step right after light launches, 36-37 after 1 s of
        # full load, 40 after 3 s of it, 42 after 20 ms of idling): a short ramp, then the untimed steps (one launch per
        # step: light), then the timed launches with nothing but the contract's barrier + synchronize in between.
        h.set_option("persistent", 0)
        h.set_option("keep_predictor", 1)
        pre_states = h.new_state(B, tdt, n_slots=pre + 1) if pre else None
        states = h.new_state(B, tdt, n_slots=3)
        ramp(B, RAMP_B)  # (on the second handle: the predictor image of `h` is not touched)
        if pre:
            h.init_straight(pre_states[0])
            h.simulate(ctl_pre, pre_states, G, tip=tip_pre)
            states[0].copy_(pre_states[pre])
            prev_init = pre_states[pre - 1].clone()
        else:
            h.init_straight(states[0])
            prev_init = None
        h.set_option("persistent", persistent_default)
        walls, kernels = [], []
        for c in range(n_chunks):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t_start = time.perf_counter()
            ev0.record()
            h.simulate(ctl_k[c], states, G, ring=True, tip=tip[c], status=status[c], prev_init=prev_init)
            ev1.record()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t_start
            if world > 1:
                dist.barrier()
                tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                elapsed = float(tmax.item())
            walls.append(elapsed)
            kernels.append(ev0.elapsed_time(ev1))  # duration of the K-step region on the launch stream (HIP events)
            if c + 1 < n_chunks:
                # the ring leaves the newest state in slot K % 3 and the one before it in (K - 1) % 3; the next call
                # starts from slot 0 again (untimed device copies)
                newest, older = states[K % 3].clone(), states[(K - 1) % 3].clone()
                states[0].copy_(newest)
                prev_init = older
        h.set_option("keep_predictor", 0)
        return {"B": B, "walls": walls, "kernel_ms": kernels, "n_bad": int((status != 0).sum()), "cold": cold,
                "tip": tip, "tip_pre": tip_pre, "path": h.get_option("last_sim_path"),
                "kernel": bl.kernel_label(h), "waves_per_rod": h.get_option("last_waves_per_rod")}

    strong = args.scaling == "strong"
    if strong and args.batch % world:
        raise SystemExit("--scaling strong needs --batch divisible by the number of GPUs")
    B = args.batch // world if strong else args.batch
    m = headline(B, world, rank, with_cold=not args.no_cpu)

    out = None
    if rank == 0:
        walls, kms = m["walls"], m["kernel_ms"]
        elapsed = statistics.median(walls)
        kernel_ms = kms[walls.index(elapsed)] if elapsed in walls else statistics.median(kms)
        rod_steps = world * B * K
        path = m["path"]  # 2: the K steps of a chunk ran as one persistent launch
        persistent = path == 2
        launches = 1 if persistent else K
        units_per_launch = B * (K if persistent else 1)  # rod-steps one launch processes
        kernel_ms = kernel_ms / launches
        # HBM, priced for the mode that RAN.  The timed calls are tip-only runs on a 3-slot ring: algorithmically a
        # rod-step must read its 4 tensions and write its tip, (3 + 4) s; what the kernel additionally
        # parks in HBM for a roll-back (the twelve leading slots of interior states, counter-measured below) is its
        # own choice.  The full-trajectory mode ((25 N + 4) s per rod-step) is measured by extra.headline_full_trajectory.
        per_rod_step = (3 + 4) * esize if persistent else (75 * N + 16) * esize
        alg_bytes = units_per_launch * per_rod_step
        hbm_achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        # SURVEY 8d algorithmic flops per rod-step: (N-1) (7k+1) F_ode with k = 3 Newton iterations, F_ode = 450
        flops_per_rod_step = (N - 1) * 22 * 450
        tf_achieved = units_per_launch * flops_per_rod_step / (kernel_ms * 1e-3) / 1e12
        prof_hbm = committed_profile(B, N, args.dtype, path, "hbm")
        prof_sq = committed_profile(B, N, args.dtype, path, "sq")
        traffic = None
        measured_per_unit = None
        if prof_hbm:
            measured_per_unit = prof_hbm["hbm_bytes_per_launch_corrected"] / (B * prof_hbm.get("steps_per_launch", 1))
            traffic = int(measured_per_unit * units_per_launch)
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
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "timed_chunks": {
                "n": n_chunks, "steps_each": K,
                "what": "consecutive K-step chunks of ONE trajectory per rod, each its own kr_simulate_batch call bracketed "
                        "by barrier + synchronize (MAX over ranks); value / ms_per_step = the median chunk",
                "value_min": round(rod_steps / max(walls), 1), "value_median": round(rod_steps / elapsed, 1),
                "value_max": round(rod_steps / min(walls), 1), "value_first": round(rod_steps / walls[0], 1),
                "wall_ms": [round(w * 1e3, 4) for w in walls], "kernel_ms": [round(k, 4) for k in kms],
            },
            "config": {
                "workload": f"forward simulate, B={B} rods/GPU, N={N}, Euler shooting + BDF2, NN off, "
                            f"setup_robot(None), per-rod sine tensions rng({SEED})",
                "rods_per_gpu": B, "N": N, "unconverged_rod_steps": m["n_bad"],
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
                "executed": executed_roofline("headline", units_per_launch / (kernel_ms * 1e-3), args.dtype),
                "valu_issue_frac": valu_issue,
                "hbm": {"achieved": round(hbm_achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(hbm_achieved / HBM_PEAK_GBS, 7), "algorithmic_bytes_per_launch": alg_bytes,
                        "algorithmic_bytes_per_rod_step": per_rod_step,
                        "mode": "tip-only run on a 3-slot ring: (3 + 4) s per rod-step" if persistent else
                                "one launch per step: (75 N + 16) s per rod-step",
                        "measured_bytes_per_rod_step": measured_per_unit and round(measured_per_unit, 1),
                        "measured_frac": measured_per_unit and round(
                            measured_per_unit * units_per_launch / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "full_trajectory_mode": "extra.headline_full_trajectory"},
                "kernel": m["kernel"],
                "kernel_ms": round(kernel_ms, 4),
                "launches": launches,
                "profile": {"hbm": prof_hbm and prof_hbm["_file"], "sq": prof_sq and prof_sq["_file"]},
                "note": "one rod per wavefront on each of the 1024 SIMDs: the launch is bound by the fp64 vector "
                        "issue rate of a single wave, not by HBM; `achieved` prices SURVEY 8d's algorithmic flops per "
                        "rod-step against the fp64 vector peak, `valu_issue_frac` is measured VALU issue (SQ_INSTS_VALU "
                        "x 4 cycles over SIMD-cycles, profiles/*pmc_sq.json), `hbm` the bytes of the mode that ran",
            },
        }
        if m["cold"] is not None:
            out["cold_start"] = {"unit": "rod-steps/s", **m["cold"],
                                 "note": "T steps from the straight rod in one call, no warm-up, no predictor hand-over "
                                         "(SURVEY 8d cfg3: T=64, cfg2: T=200), best of 3"}
        if cpu_leg is not None:
            Tc, cb, tips, cbc = cpu_leg
            # tip parity of the TIMED batch against the oracle: same rods, same steps (untimed + first timed ones)
            tip_all = torch.cat([m["tip_pre"][:, :pre]] + [m["tip"][c] for c in range(n_chunks)], dim=1)
            gpu_tips = tip_all[: len(tips), :Tc].double().cpu().numpy()
            errs = [float(np.linalg.norm(gpu_tips[b] - tips[b]) / np.linalg.norm(tips[b])) for b in range(len(tips))]
            out["tip_rel_l2_vs_oracle"] = max(errs)
            out["tip_check"] = {"rods": len(tips), "steps": Tc, "timed_steps_included": max(0, Tc - pre),
                                "what": "rods 0.. of the timed batch, steps 1..steps of their trajectory"}
            out["cpu_baseline"] = cb
            out["cpu_baseline_c"] = cbc

    extra = {}
    if not args.no_extra:
        if world == 1:
            extra.update(run_extras(torch, dev_index, refs, log))
        else:
            # (a) strong scaling beside the default weak mode (or the other way round): the same global draw of rods
            other = "weak" if strong else "strong"
            if args.batch % world == 0:
                B2 = args.batch if strong else args.batch // world
                m2 = headline(B2, world, rank, with_cold=False)
                if rank == 0:
                    w2 = statistics.median(m2["walls"])
                    extra[f"{other}_scaling"] = {
                        "value": round(world * B2 * K / w2, 1), "unit": "rod-steps/s", "rods_per_gpu": B2,
                        "ms_per_step": round(w2 / K * 1e3, 4), "kernel": m2["kernel"], "waves_per_rod": m2["waves_per_rod"],
                        "unconverged": m2["n_bad"], "wall_ms": [round(w * 1e3, 4) for w in m2["walls"]]}
            # (b) BASELINE cfg4: data-parallel training with the gradient all-reduce (RCCL over xGMI)
            try:
                dp = bl.train_dp_leg(torch, dist, dev_index, world, rank, backend)
            except Exception as e:
                dp = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                extra["train_dp"] = dp
    if rank == 0:
        if extra:
            out["extra"] = extra
            out["extra_keys"] = sorted(extra)
        print(json.dumps(out), flush=True)
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
    avail = len(os.sched_getaffinity(0))
    cores = min(avail, CPU_WORKERS_MAX)
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
    return {"value": round(rods * steps / t_all, 1), "unit": "rod-steps/s", "cores": cores, "cores_available": avail,
            "kind": "port", "sample": f"{rods} rods x {steps} steps of the bench workload (N={N}, fp64), scalar C, Newton shooting to 1e-12, "
                      f"one rod per thread", "single_core_value": round(steps / t_single, 1),
            "unconverged": int(sum(r[2] for r in res) + one[2])}


if __name__ == "__main__":
    main()
