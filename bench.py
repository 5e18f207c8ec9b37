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

N>1: launched by torch.distributed.run, one rank per GPU; rods are sharded
(weak scaling, no data-path collective); the only collectives are the timing
barrier and the MAX over ranks of the elapsed time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def pmc_traffic(B, N, dtype, path, steps):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/*pmc_hbm.json,
    FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE); None if no profile matches this workload."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_hbm.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == f"B={B} N={N} {dtype} Euler" and d.get("sim_path", 0) == path and d.get("steps_per_launch", 1) == steps:
            best = d.get("hbm_bytes_per_launch_corrected")
    return best


def cpu_baseline(N, del_t_unused, sample_steps=50):
    """Times the oracle (NumPy port of cosserat_ode.py + knode.simulate with
    scipy fsolve - the reference's own execution model) on the host: one rod
    per process on every available core, same workload definition."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import multiprocessing as mp
    cores = min(len(os.sched_getaffinity(0)), 16)
    t0 = time.perf_counter()
    single = _cpu_worker((N, 0, sample_steps))
    t_single = time.perf_counter() - t0
    rods = 2 * cores  # >= 32 rods x 50 steps on all host cores
    with mp.get_context("fork").Pool(cores) as pool:
        t0 = time.perf_counter()
        tips = pool.map(_cpu_worker, [(N, b, sample_steps) for b in range(rods)], chunksize=1)
        t_all = time.perf_counter() - t0
    return {
        "value": round(rods * sample_steps / t_all, 3),
        "unit": "rod-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{rods} rods x {sample_steps} steps of the bench workload (N={N}, fp64, fsolve shooting), "
                  f"one rod per process, {cores} processes",
        "single_core_value": round(sample_steps / t_single, 3),
    }, tips


def _cpu_worker(args):
    N, b, steps = args
    os.environ["OMP_NUM_THREADS"] = "1"
    import numpy as np
    import cosserat_oracle as orc
    D = orc.params_for(None, N).derived()
    ctl = orc.batch_sine_controls(max(b + 1, 16), steps, D.P.del_t, 1235)[b]
    # the oracle mirrors knode.simulate: T controls -> T solves, last one dropped from the output
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
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    args = ap.parse_args()

    import numpy as np
    import torch
    import krod_native as kn
    from cosserat_ode import CosseratRod
    from knode import setup_robot

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
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
            dist.init_process_group("nccl", device_id=torch.device(dev))
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

    # synthetic inputs, SURVEY 8d: rod b of the global batch gets its own period and phase
    rng = np.random.default_rng(1235)
    Pd = rng.uniform(0.5, 3.0, size=B * world)[rank * B:(rank + 1) * B]
    phi = rng.uniform(0.0, 2 * np.pi, size=B * world)[rank * B:(rank + 1) * B]
    k = np.arange(4)[None, None, :]

    # state lives in HBM as a 3-slot ring of packed states; kr_simulate_batch advances it W (+K) steps.
    # W is rounded up to a multiple of 3 so that the timed call starts at ring slot 0 again, with the
    # state before it in slot 2 (handed over as state_prev_init: the run continues exactly).
    W = (W + 2) // 3 * 3
    # the start-value predictor needs about 20 steps of history to settle: never fewer
    # warm-up steps than that, whatever was asked for; the JSON line reports the number actually done
    W = max(W, 30)
    i = np.arange(1, W + K + 1)[None, :, None]
    ctl = 6.0 + np.sin(2 * np.pi * i * robot.del_t / Pd[:, None, None] + phi[:, None, None] + k * (np.pi / 2))
    ctl_w = torch.as_tensor(ctl[:, :W], device=dev).to(tdt).contiguous()
    ctl_k = torch.as_tensor(ctl[:, W:], device=dev).to(tdt).contiguous()
    states = h.new_state(B, tdt, n_slots=3)
    h.init_straight(states[0])
    G = torch.zeros((B, 6), dtype=tdt, device=dev)
    status = torch.zeros((B, K), dtype=torch.int32, device=dev)
    tip = torch.empty((B, K, 3), dtype=tdt, device=dev)

    # clock ramp, not part of the W warm-up steps: an idle MI355X needs a few hundred ms of load before
    # its shader clock settles; the same kernel runs on a scratch copy of the problem until then
    # (ramp and warm-up steps go through the one-launch-per-step form of the same solver, so that the
    # persistent kernel appears in a rocprofv3 trace exactly once: the timed K steps)
    persistent_default = h.get_option("persistent")
    h.set_option("persistent", 0)
    scratch = h.new_state(B, tdt, n_slots=3)
    Gs = torch.zeros((B, 6), dtype=tdt, device=dev)
    ctl_r = (ctl_w if W else ctl_k[:, :30]).contiguous()
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.5:
        h.init_straight(scratch[0])
        Gs.zero_()
        h.simulate(ctl_r, scratch, Gs, ring=True)
        torch.cuda.synchronize()
    del scratch

    # the W warm-up steps and the K timed steps are one trajectory advanced by two calls: the second call
    # resumes the start-value predictor of the first (option "keep_predictor"), so that a short timed region
    # is as representative of the steady state as a long one
    h.set_option("keep_predictor", 0)
    h.set_option("keep_predictor", 1)
    if W:
        h.simulate(ctl_w, states, G, ring=True)
    h.set_option("persistent", persistent_default)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    ev0.record()
    h.simulate(ctl_k, states, G, ring=True, tip=tip, status=status, prev_init=states[2] if W else None)
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_ms = ev0.elapsed_time(ev1)  # duration of the K-step region on the launch stream (HIP events)

    n_bad = int((status != 0).sum())

    if rank == 0:
        rod_steps = world * B * K
        path = h.get_option("last_sim_path")  # 2: the K steps ran as one persistent launch
        persistent = path == 2
        # SURVEY 8d algorithmic bytes per rod-step: state written every step (25 N + 4) s in the persistent
        # form (history never leaves the CU); (75 N + 16) s when every step is its own launch
        per_rod_step = (25 * N + 4) * esize if persistent else (75 * N + 16) * esize
        launches = 1 if persistent else K
        alg_bytes = B * per_rod_step * (K if persistent else 1)
        kernel_ms = kernel_ms / launches
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "rod-steps/sec (N=100 segments, batch=1024)",
            "value": round(rod_steps / elapsed, 1),
            "unit": "rod-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"forward simulate, B={B} rods/GPU, N={N}, Euler shooting + BDF2, NN off, "
                            f"setup_robot(None), per-rod sine tensions rng(1235)",
                "rods_per_gpu": B, "N": N, "unconverged_rod_steps": n_bad,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": pmc_traffic(B, N, args.dtype, path, K if persistent else 1),
                "kernel": ("kr::ms_sim_kernel (persistent, all K steps in one launch)", "kr::ms_step_kernel",
                           "kr::step_kernel")[2 - path] if path in (0, 1, 2) else "?",
                "kernel_ms": round(kernel_ms, 4),
                "launches": launches,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "fp64 VALU issue bound, not HBM bound: one rod per wavefront on each of the 1024 SIMDs, "
                        "2 Newton sweeps x 25 grid points x ~180 fp64 instructions per step; VALU busy ~58% of wave cycles "
                        "(profiles/*pmc_sq.json, DESIGN.md section 4)",
            },
        }
        if world == 1 and not args.no_cpu:
            cb, tips = cpu_baseline(N, robot.del_t)
            # tip parity of the GPU run against the oracle on the rods the CPU leg simulated (first steps)
            import numpy as _np
            Tc = tips[0].shape[0]
            if Tc <= W + K:
                st2 = h.new_state(len(tips), tdt, n_slots=3)
                h.init_straight(st2[0])
                G2 = torch.zeros((len(tips), 6), dtype=tdt, device=dev)
                tip2 = torch.empty((len(tips), Tc, 3), dtype=tdt, device=dev)
                c2 = torch.as_tensor(_np.stack([_cpu_ctl(b, Tc, robot.del_t) for b in range(len(tips))]), device=dev).to(tdt)
                h.simulate(c2.contiguous(), st2, G2, ring=True, tip=tip2)
                torch.cuda.synchronize()
                g = tip2.cpu().numpy()
                errs = [float(_np.linalg.norm(g[b] - tips[b]) / _np.linalg.norm(tips[b])) for b in range(len(tips))]
                out["tip_rel_l2_vs_oracle"] = max(errs)
            out["cpu_baseline"] = cb
            try:
                out["cpu_baseline_c"] = cpu_baseline_c(N, robot.del_t)
            except Exception as e:  # the C restatement is optional test infrastructure (needs gcc or its prebuilt .so)
                out["cpu_baseline_c"] = {"error": str(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline_c(N, del_t, rods=256, steps=150):
    """The scalar C restatement (oracle/cosserat_oracle_c.c, Newton shooting) on all host cores, one rod per call,
    threads (the C call releases the GIL): what an optimised CPU implementation of the same discrete equations
    reaches, beside the NumPy port that has the reference's own execution model."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from concurrent.futures import ThreadPoolExecutor
    import cosserat_oracle as orc
    import cosserat_oracle_c as oc
    cores = min(len(os.sched_getaffinity(0)), 16)
    P = orc.params_for(None, N)
    ctl = orc.batch_sine_controls(rods, steps, del_t, 1235)
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


def _cpu_ctl(b, steps, del_t):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cosserat_oracle as orc
    return orc.batch_sine_controls(max(b + 1, 16), steps, del_t, 1235)[b]


if __name__ == "__main__":
    main()
