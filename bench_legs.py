"""The legs of bench.py beside the headline: the other BASELINE.json configurations, measured by the same driver
run (one JSON sub-object each under ``extra``), and the data-parallel training leg that puts the RCCL all-reduce of
SURVEY 8e on the driver's command line (``--gpus N > 1``).

Nothing here imports ``oracle/``: reference tips for the accuracy fields are computed by bench.py's CPU leg (before the
first GPU call) and handed in.  Inputs follow SURVEY 8d: per-rod sine tensions from ``default_rng(seed)``, MLP weights
|N(0.01, 0.01)|, biases N(0, 0.01) from ``default_rng(7)`` (cosserat_ode_torch.py:76-105)."""
import os
import statistics
import time

import numpy as np

FP64_PEAK_TF = 78.6      # fp64 vector = fp64 matrix peak of an MI355X (public figure)
FP32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: fp32 vector = fp32 MFMA peak
HBM_PEAK_GBS = 8000.0
F_ODE = 450              # flops of one ODE evaluation incl. the Euler update
FD_EVALS = 22            # 7 k + 1 evaluations per grid point with k = 3 Newton iterations


def sine_controls(B, steps, del_t, seed, world=1, rank=0, first_step=1):
    """Tensions [B, steps, 4] of this rank's rods: rod b of the GLOBAL batch of world x B rods has period
    P_b ~ U[0.5, 3] s and phase phi_b ~ U[0, 2 pi) from default_rng(seed); rank r owns rods [r B, (r + 1) B).
    Step i (1-based from the straight rod) applies 6 + sin(2 pi i dt / P_b + phi_b + k pi / 2)."""
    rng = np.random.default_rng(seed)
    Pd = rng.uniform(0.5, 3.0, size=B * world)[rank * B:(rank + 1) * B]
    phi = rng.uniform(0.0, 2 * np.pi, size=B * world)[rank * B:(rank + 1) * B]
    k = np.arange(4)[None, None, :]
    i = np.arange(first_step, first_step + steps)[None, :, None]
    return 6.0 + np.sin(2 * np.pi * i * del_t / Pd[:, None, None] + phi[:, None, None] + k * (np.pi / 2))


def mlp_weights(sizes, seed=7):
    """(weights, biases) of a dense stack, drawn like the reference initialises its networks."""
    rng = np.random.default_rng(seed)
    Ws, bs = [], []
    for k in range(len(sizes) - 1):
        Ws.append(np.abs(rng.normal(0.01, 0.01, size=(sizes[k + 1], sizes[k]))).astype(np.float32))
        bs.append(rng.normal(0.0, 0.01, size=(sizes[k + 1],)).astype(np.float32))
    return Ws, bs


def inject_mlp(robot, Ws, bs):
    """What physics_train.py:104-110 does to switch the NumPy rod to NN mode (ELU between the layers)."""
    model, params = [], []
    for k, (W, b) in enumerate(zip(Ws, bs)):
        model.append(f"Linear(in_features={W.shape[1]}, out_features={W.shape[0]}, bias=True)")
        params += [W, b]
        if k < len(Ws) - 1:
            model.append("ELU(alpha=1.0)")
    robot.nn_model, robot.param_ls, robot.nn_path = model, params, "bench"


def kernel_label(h):
    path = h.get_option("last_sim_path")
    wpr = h.get_option("last_waves_per_rod")
    if wpr > 1:
        if path == 2 and h.get_option("last_overlap"):
            return f"kr::mswo_sim_kernel (persistent, overlapped steps, {wpr} wavefronts per rod)"
        return (f"kr::msw_sim_kernel (persistent, {wpr} wavefronts per rod)" if path == 2
                else f"kr::msw_step_kernel ({wpr} wavefronts per rod)")
    if path == 2 and h.get_option("last_overlap"):
        return "kr::mso_sim_kernel (persistent, overlapped steps)"
    return {0: "kr::step_kernel", 1: "kr::ms_step_kernel", 2: "kr::ms_sim_kernel (persistent)"}.get(path, "?")


def make_robot(N, dev_index, mod=None):
    from cosserat_ode import CosseratRod
    from knode import setup_robot
    r = CosseratRod(use_fsolve=True, device=dev_index)
    setup_robot(r, mod)
    r.N = N
    r.compute_intermediate_terms()
    return r


def _rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def forward_leg(torch, dev_index, B, N, T, warm, dtype, seed, mlp=None, full_trajectory=False, ref_tips=None,
                repeats=3):
    """`warm` untimed + T timed steps of one trajectory per rod (two calls, predictor handed over), best of `repeats`;
    then - for the accuracy field - the first steps from the straight rod against the oracle's tips of rods 0, 1."""
    import krod_native as kn  # noqa: F401
    dev = f"cuda:{dev_index}"
    tdt = torch.float64 if dtype == "f64" else torch.float32
    esize = 8 if dtype == "f64" else 4
    r = make_robot(N, dev_index)
    if mlp is not None:
        inject_mlp(r, *mlp)
    h = r._native()
    use_nn = mlp is not None
    ctl = torch.as_tensor(sine_controls(B, warm + T, r.del_t, seed), device=dev).to(tdt).contiguous()
    ctl_w = ctl[:, :warm].contiguous() if warm else None
    ctl_t = ctl[:, warm:].contiguous()
    G = torch.zeros((B, 6), dtype=tdt, device=dev)
    status = torch.zeros((B, T), dtype=torch.int32, device=dev)
    h.set_option("keep_predictor", 1 if warm else 0)
    n_slots = (T + 1) if full_trajectory else 3
    st = h.new_state(B, tdt, n_slots=n_slots)
    st_w = h.new_state(B, tdt, n_slots=3) if (warm and full_trajectory) else None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best_wall, best_ev = 1e30, 1e30
    for _ in range(repeats):
        G.zero_()
        prev_init = None
        if warm:
            sw = st_w if full_trajectory else st
            h.init_straight(sw[0])
            h.simulate(ctl_w, sw, G, ring=True, use_nn=use_nn)
            # the ring leaves the newest state in slot warm % 3, the one before it in (warm - 1) % 3
            newest, older = sw[warm % 3], sw[(warm - 1) % 3]
            if full_trajectory:
                st[0].copy_(newest)
                prev_init = older.clone()
            else:
                prev_init = older.clone()
                if warm % 3:
                    tmp = newest.clone()
                    st[0].copy_(tmp)
        else:
            h.init_straight(st[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        h.simulate(ctl_t, st, G, ring=not full_trajectory, status=status, use_nn=use_nn, prev_init=prev_init)
        e1.record()
        torch.cuda.synchronize()
        best_wall = min(best_wall, time.perf_counter() - t0)
        best_ev = min(best_ev, e0.elapsed_time(e1) * 1e-3)
    h.set_option("keep_predictor", 0)
    out = {
        "value": round(B * T / best_wall, 1), "unit": "rod-steps/s", "ms_per_step": round(best_wall / T * 1e3, 5),
        "kernel_ms_per_step": round(best_ev / T * 1e3, 5),
        "B": B, "N": N, "T": T, "untimed_steps": warm, "dtype": dtype, "output": "full trajectory" if full_trajectory else "tips (3-slot ring)",
        "kernel": kernel_label(h), "path": h.get_option("last_sim_path"), "waves_per_rod": h.get_option("last_waves_per_rod"),
        "unconverged": int((status != 0).sum()), "repeats": repeats,
    }
    # roofline (nominal): physics flops per rod-step, + the MLP's if it is on; HBM for the stored states
    flops = (N - 1) * FD_EVALS * F_ODE
    how = f"(N-1) x {FD_EVALS} x {F_ODE} flop per rod-step (SURVEY 8d, k = 3 FD-Newton iterations)"
    bound = "valu_fp64" if dtype == "f64" else "valu_fp32"
    if mlp is not None:
        mac = sum(int(W.shape[0]) * int(W.shape[1]) for W in mlp[0])
        flops += (N - 1) * FD_EVALS * 2 * mac
        how += f" + (N-1) x {FD_EVALS} x 2 x {mac} flop of the MLP (nominal: the kernel evaluates the perturbed columns as JVPs)"
        bound = "mfma"
    peak = FP64_PEAK_TF if dtype == "f64" else FP32_PEAK_TF
    tf = B * T * flops / best_ev / 1e12
    out["roofline"] = {"bound": bound, "achieved": round(tf, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 5),
                       "how": how}
    if mlp is not None:
        # NOT a utilisation: the nominal count prices 22 full evaluations of the network per grid point and sweep in the
        # arithmetic type of the run, the kernel evaluates 4 rows that way and the perturbed columns as bf16 JVPs (and, on
        # storing sweeps, the 4 rows only) - the ratio says how far the method is from the naive operation count
        out["roofline"]["frac"] = None
        out["roofline"]["nominal_over_vector_peak"] = round(tf / peak, 5)
        out["roofline"]["note"] = ("nominal operation count (every forward-difference column a full network evaluation) over the "
                                   "vector peak of the run's type - not a utilisation figure: the kernel runs the columns as bf16 JVPs "
                                   "on the matrix cores; executed matrix / vector instruction counts: profiles/*_kernels.json "
                                   "(simulate_nn_*: SQ_INSTS_MFMA, SQ_INSTS_VALU)")
    per_rod_step = (25 * N + 4) * esize if full_trajectory else (3 + 4) * esize
    gbs = B * T * per_rod_step / best_ev / 1e9
    out["roofline"]["hbm"] = {"achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 6),
                              "algorithmic_bytes_per_rod_step": per_rod_step,
                              "how": "(25 N + 4) s: every state leaves the chip once" if full_trajectory
                              else "(3 + 4) s: tip out, tensions in (tip-only run; the kernel additionally parks what a roll-back needs)"}
    # accuracy: the same batch from the straight rod, rods 0..1 against the oracle
    if ref_tips is not None:
        Tc = ref_tips.shape[1]
        c = torch.as_tensor(sine_controls(B, Tc, r.del_t, seed), device=dev).to(tdt).contiguous()
        s2 = h.new_state(B, tdt, n_slots=3)
        h.init_straight(s2[0])
        g2 = torch.zeros((B, 6), dtype=tdt, device=dev)
        tip = torch.empty((B, Tc, 3), dtype=tdt, device=dev)
        h.simulate(c, s2, g2, ring=True, tip=tip, use_nn=use_nn)
        torch.cuda.synchronize()
        got = tip[: ref_tips.shape[0]].double().cpu().numpy()
        out["tip_rel_l2_vs_oracle"] = max(_rel(got[b], ref_tips[b]) for b in range(ref_tips.shape[0]))
        out["tip_check"] = f"rods 0..{ref_tips.shape[0] - 1} of this batch, steps 1..{Tc} from the straight rod, same kernel ({kernel_label(h)})"
    else:
        out["tip_rel_l2_vs_oracle"] = None
    return out


def device_trajectories(torch, robot, ctl_np):
    """float32 [M, T, 25, N] on the device: what knode.simulate(robot, ctl)[:, :25] returns for every row of ctl
    (entry 0 = the straight rod, the T-th solve dropped: knode.py:102), without a host round trip."""
    h = robot._native()
    dev = f"cuda:{robot.device}"
    M, T = ctl_np.shape[0], ctl_np.shape[1]
    ctl = torch.as_tensor(ctl_np[:, : T - 1], device=dev).float().contiguous()
    st = h.new_state(M, torch.float32, n_slots=T)
    h.init_straight(st[0])
    G = torch.zeros((M, 6), dtype=torch.float32, device=dev)
    status = torch.zeros((M, T - 1), dtype=torch.int32, device=dev)
    h.simulate(ctl, st, G, status=status)
    traj = torch.empty((M, T, 25, h.N), dtype=torch.float32, device=dev)
    for t in range(T):
        y, z = h.unpack(st[t])
        traj[:, t, :19] = y
        traj[:, t, 19:] = z
    return traj, int((status != 0).sum())


def torch_rod(torch, dev, N, layers, seed=7, mod="damping"):
    """The trainable rod of physics_train.py:182 with an imperfect model (`mod`) and an MLP 28 -> layers -> 25 (ELU)
    whose initial weights come from NumPy (the same on every rank)."""
    import torch.nn as nn
    from cosserat_ode_torch import CosseratRodTorch
    from knode import setup_robot
    rob = CosseratRodTorch(dev, layers[0])
    setup_robot(rob, mod)
    rob.N = N
    rob.compute_intermediate_terms()
    sizes = [28] + list(layers) + [25]
    Ws, bs = mlp_weights(sizes, seed)
    mods = []
    for k, (W, b) in enumerate(zip(Ws, bs)):
        lin = nn.Linear(W.shape[1], W.shape[0])
        with torch.no_grad():
            lin.weight.copy_(torch.as_tensor(W))
            lin.bias.copy_(torch.as_tensor(b))
        mods.append(lin)
        if k < len(Ws) - 1:
            mods.append(nn.ELU())
    rob.nn_models = nn.ModuleList(mods).to(dev)
    return rob, sizes


def train_leg(torch, dev_index, M, T, N, key_pts, layers, ctl_seed=1236, epochs=100, ramp_epochs=300):
    """One epoch of the fused training step (forward + 4-term loss + backward + Adam + plateau schedule + clamp,
    physics_train.py:306-408) on M trajectories of T entries: median over `epochs` single-epoch timings (HIP events),
    taken after `ramp_epochs` untimed epochs - the reference's loop runs thousands of epochs, and the first ~100 of a
    process run at a lower shader clock (tools/train_clock.py: 136 us for epochs 0..19, 128 us from epoch ~100 on at
    cfg3; the headline leg ramps the clock for the same reason).  `wall_us_per_epoch`: wall clock of `epochs` epochs
    queued by ONE kr_train_epochs call; `wall_us_per_epoch_step_calls`: the same through one Python call per epoch."""
    from krod_train import KnodeTrainer
    dev = f"cuda:{dev_index}"
    rr = make_robot(N, dev_index)
    ctl = sine_controls(M, T, rr.del_t, ctl_seed)
    traj, bad = device_trajectories(torch, rr, ctl)
    rob, sizes = torch_rod(torch, dev, N, layers)
    tr = KnodeTrainer(rob, traj, torch.as_tensor(ctl, device=dev).float().contiguous(), key_pts, keep_pred=False)
    tr.run(ramp_epochs)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(epochs)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record()
        tr.step(sync_loss=False)
        b.record()
    torch.cuda.synchronize()
    wall_steps = (time.perf_counter() - t0) / epochs
    t0 = time.perf_counter()
    tr.run(epochs)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / epochs
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    med = statistics.median(us)
    flops = 6 * tr.Q * sum(a * b for a, b in zip(sizes[:-1], sizes[1:]))
    tf = flops / (med * 1e-6) / 1e12
    losses = tr.losses()
    return {
        "value": round(M * (T - 1) / (med * 1e-6), 1), "unit": "trajectory-steps/s", "us_per_epoch": round(med, 2),
        "us_per_epoch_min": round(us[0], 2), "us_per_epoch_max": round(us[-1], 2), "wall_us_per_epoch": round(wall * 1e6, 2),
        "wall_us_per_epoch_step_calls": round(wall_steps * 1e6, 2), "timed_epochs": epochs, "ramp_epochs": ramp_epochs,
        "rows": tr.Q, "network": "->".join(map(str, sizes)), "trajectories": M, "window_steps": T - 1, "key_points": list(map(int, key_pts)),
        "N": N, "dtype": "f32",
        "kernel": ("kr_train_epoch: " + ("kr::mlp_fwd3_kernel (+ loss epilogue), kr::mlp_bwd3_kernel"
                                        if len(layers) == 2 else "kr::mlp_fwd2_kernel (+ loss epilogue), kr::mlp_bwd2_kernel")
                   + ", kr::train_tail_kernel (slab + loss sums, Adam, clamp, plateau schedule, fragment update)")
                  if tr.fused_epoch else "kr::mlp_fwd_fused_kernel (+ loss epilogue), kr::mlp_bwd*, kr::adam_plateau_kernel",
        "launches_per_epoch": 3 if tr.fused_epoch else None,
        "data_unconverged": bad, "loss_first": losses[0], "loss_last": losses[-1],
        "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": FP32_PEAK_TF, "unit": "TFLOP/s",
                     "frac": round(tf / FP32_PEAK_TF, 5),
                     "how": "6 x rows x sum(in x out) useful fp32 flops (forward + both backward products) per epoch over "
                            "the median epoch time, against the fp32 MFMA peak"},
    }


def mlp_literal_leg(torch, dev_index, Q, repeats=21):
    """BASELINE.json configs[2] read literally: a KNODE residual MLP 18 -> 64 -> 64 -> 6 (ELU), forward + backward over
    Q rows as a bare GEMM micro-benchmark (SURVEY 8d cfg3 asks to report it beside the 28 -> 64 -> 64 -> 25 network of the
    reference's I/O contract, which `cfg3_train_epoch` times).  There is no loss for 6 outputs in the reference: dout is a
    fixed random matrix.  kr_mlp_forward + kr_mlp_backward through the C ABI, median of `repeats` pairs (HIP events)."""
    import ctypes as C
    import krod_native as kn
    dev = f"cuda:{dev_index}"
    h = make_robot(100, dev_index)._native()
    dims = [18, 64, 64, 6]
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    Ws = [torch.randn(dims[k + 1], dims[k], device=dev, generator=gen) * 0.1 for k in range(3)]
    bs = [torch.randn(dims[k + 1], device=dev, generator=gen) * 0.1 for k in range(3)]
    dWs = [torch.zeros_like(w) for w in Ws]
    dbs = [torch.zeros_like(b) for b in bs]
    x = torch.zeros((Q, 32), device=dev)
    x[:, :18] = torch.randn(Q, 18, device=dev, generator=gen)
    out = torch.zeros((Q, 32), device=dev)
    dout = torch.zeros((Q, 32), device=dev)
    dout[:, :6] = torch.randn(Q, 6, device=dev, generator=gen)
    dims_c = (C.c_int32 * 4)(*dims)
    acts_c = (C.c_int32 * 3)(kn.ACT_ELU, kn.ACT_ELU, kn.ACT_NONE)
    ws = torch.empty(max(h.lib.kr_mlp_ws_bytes(3, dims_c, Q), 16), dtype=torch.uint8, device=dev)
    Wp = (C.c_void_p * 3)(*[w.data_ptr() for w in Ws])
    bp = (C.c_void_p * 3)(*[b.data_ptr() for b in bs])
    dWp = (C.c_void_p * 3)(*[w.data_ptr() for w in dWs])
    dbp = (C.c_void_p * 3)(*[b.data_ptr() for b in dbs])

    def fb():
        kn.check(h.lib.kr_mlp_forward(h._h, Q, 3, dims_c, acts_c, Wp, bp, kn._ptr(x), 32, kn._ptr(out), kn._ptr(ws), kn._stream()))
        kn.check(h.lib.kr_mlp_backward(h._h, Q, 3, dims_c, acts_c, Wp, kn._ptr(x), 32, kn._ptr(dout), kn._ptr(ws), dWp, dbp,
                                       kn._stream()))
    for _ in range(200):  # (first use of every kernel, then the same clock ramp as the training legs: tools/train_clock.py)
        fb()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(repeats)]
    for a, b in evs:
        a.record()
        fb()
        b.record()
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    med = statistics.median(us)
    # parity of this very call against fp64 torch on a slice (the micro-benchmark must not time a wrong answer)
    n_chk = 4096
    import torch.nn.functional as F
    a = x[:n_chk, :18].double()
    for k in range(3):
        a = a @ Ws[k].double().t() + bs[k].double()
        if k < 2:
            a = F.elu(a)
    err = float((out[:n_chk, :6].double() - a).norm() / a.norm())
    mac = sum(p * q for p, q in zip(dims[:-1], dims[1:]))
    tf = 6 * Q * mac / (med * 1e-6) / 1e12
    return {"value": round(Q / (med * 1e-6), 1), "unit": "rows/s", "us_per_fwd_bwd": round(med, 2), "us_min": round(us[0], 2),
            "us_max": round(us[-1], 2), "rows": Q, "network": "18->64->64->6 (BASELINE-literal), ELU", "dtype": "f32",
            "kernel": "kr_mlp_forward + kr_mlp_backward (kr::pack_all_kernel, kr::mlp_fwd3_kernel, kr::mlp_bwd3_kernel, kr::reduce_slabs_kernel)",
            "padding_note": "the kernels run the padded shape 32 -> 64 -> 64 -> 32 (MFMA tiles of 16): 5 632 useful of 8 192 issued "
                            "multiply-adds per row forward, against 7 488 of 8 192 for 28 -> 64 -> 64 -> 25",
            "forward_rel_l2_vs_fp64_torch": err,
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": FP32_PEAK_TF, "unit": "TFLOP/s",
                         "frac": round(tf / FP32_PEAK_TF, 5),
                         "how": "6 x rows x sum(in x out) useful fp32 flops (forward + both backward products) over the median "
                                "forward + backward time, against the fp32 MFMA peak"}}


def tolerance_sweep_leg(torch, dev_index):
    """BASELINE cfg5 "fp64 vs fp32 tolerance sweep": the reference's own N = 400 run (tests/golden/sim_n400.npz, produced by
    the unmodified reference) re-simulated at four Newton tolerances per arithmetic type; tip rel L2 against the reference."""
    dev = f"cuda:{dev_index}"
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "sim_n400.npz"))
    r = make_robot(400, dev_index)
    h = r._native()
    out = {"fixture": "tests/golden/sim_n400.npz (reference run: one rod, N = 400, sine tensions)", "unit": "tip rel L2 vs reference",
           "steps": int(g["ctl"].shape[0])}
    t0 = time.perf_counter()
    for dname, dt, tols in (("f64", torch.float64, (1e-6, 1e-8, 1e-10, 1e-12)), ("f32", torch.float32, (1e-3, 1e-4, 1e-5, 1e-6))):
        row = {}
        for tol in tols:
            c = torch.as_tensor(g["ctl"][None], device=dev).to(dt).contiguous()
            T = c.shape[1]
            st = h.new_state(1, dt, n_slots=T + 1)
            h.init_straight(st[0])
            G = torch.zeros((1, 6), dtype=dt, device=dev)
            tip = torch.empty((1, T, 3), dtype=dt, device=dev)
            status = torch.zeros((1, T), dtype=torch.int32, device=dev)
            h.simulate(c, st, G, tip=tip, status=status, tol=tol)
            torch.cuda.synchronize()
            # the reference drops its last solve and lists the initial tip first (knode.py:96-102)
            got = np.concatenate([st[0][0, -1, 12:15].cpu().numpy()[None], tip[0].cpu().numpy()])[:T].astype(np.float64)
            row[f"{tol:.0e}"] = {"tip_rel_l2": _rel(got, g["tip"]), "unconverged": int((status != 0).sum())}
        out[dname] = row
    out["value"] = out["f64"]["1e-08"]["tip_rel_l2"]
    out["seconds"] = round(time.perf_counter() - t0, 3)
    out["contract"] = "tip rel L2 <= 1e-5 (BASELINE north_star): met by every fp64 tolerance and by fp32 at tol <= 1e-5"
    return out


def train_dp_leg(torch, dist, dev_index, world, rank, backend, M_total=4096, T=30, N=10, key_pts=(3, 5, 7, 9),
                 layers=(512,), epochs=50, check=True):
    """BASELINE cfg4: the training loop of physics_train.py:306-408 on M_total trajectories sharded over the ranks
    (SURVEY 8e): per epoch local forward / loss / backward, ONE all-reduce (SUM) of the flat gradient + loss buffer
    (RCCL over xGMI with the nccl backend), identical Adam + clamp + plateau schedule on every rank.  Rank 0 repeats
    the run alone on the whole batch; the loss curves must agree (same terms, summed in a different order)."""
    from krod_train import KnodeTrainer, shard_range
    dev = f"cuda:{dev_index}"
    rr = make_robot(N, dev_index)
    # every rank draws the same global set of tensions (sine, period U[0.5, 3] s, random phase) and keeps its shard
    ctl = sine_controls(M_total, T, rr.del_t, 1236)
    lo, hi = shard_range(M_total, rank, world)
    traj, bad = device_trajectories(torch, rr, ctl[lo:hi])
    rob, sizes = torch_rod(torch, dev, N, list(layers))
    tr = KnodeTrainer(rob, traj, torch.as_tensor(ctl[lo:hi], device=dev).float().contiguous(), list(key_pts))
    tr.time_allreduce = True
    for _ in range(2):          # first use of every kernel and of the communicator
        tr.step(sync_loss=False)
    tr.allreduce_events.clear()
    # restart from the initial weights so that the curve is the one the single-rank run produces
    rob2, _ = torch_rod(torch, dev, N, list(layers))
    tr = KnodeTrainer(rob2, traj, torch.as_tensor(ctl[lo:hi], device=dev).float().contiguous(), list(key_pts))
    tr.time_allreduce = True
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        tr.step(sync_loss=False)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        cdev = dev if backend == "nccl" else "cpu"
        tmax = torch.tensor([wall], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())
        seen = torch.ones(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        ranks_seen = int(seen.item())
    else:
        ranks_seen = 1
    ar_us = sorted(a.elapsed_time(b) * 1e3 for a, b in tr.allreduce_events)
    # the same shard WITHOUT the collective (group=False): what the all-reduce and its stream hand-off add to an epoch
    rob3, _ = torch_rod(torch, dev, N, list(layers))
    local = KnodeTrainer(rob3, traj, torch.as_tensor(ctl[lo:hi], device=dev).float().contiguous(), list(key_pts), group=False)
    for _ in range(2):
        local.step(sync_loss=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        local.step(sync_loss=False)
    torch.cuda.synchronize()
    local_us = (time.perf_counter() - t0) / epochs * 1e6
    losses = tr.losses()
    n_floats = int(tr.bucket.flat.numel())
    out = None
    if rank == 0:
        out = {
            "value": round(M_total * (T - 1) * epochs / wall, 1), "unit": "trajectory-steps/s",
            "us_per_epoch": round(wall / epochs * 1e6, 2), "epochs": epochs, "trajectories": M_total,
            "trajectories_per_rank": hi - lo, "rows_per_rank": tr.Q, "network": "->".join(map(str, sizes)), "N": N,
            "backend": backend + (" (RCCL)" if backend == "nccl" else " (rehearsal: CPU tensors)"),
            "allreduce_us": round(statistics.median(ar_us), 2) if ar_us else None,
            "allreduce_us_min_max": [round(ar_us[0], 2), round(ar_us[-1], 2)] if ar_us else None,
            "allreduce_us_stats": {"min": round(ar_us[0], 2), "median": round(statistics.median(ar_us), 2),
                                   "max": round(ar_us[-1], 2), "n": len(ar_us),
                                   "what": "HIP events around dist.all_reduce on the compute stream, rank 0"} if ar_us else None,
            "epoch_us_with_allreduce": round(wall / epochs * 1e6, 2),
            "epoch_us_without_allreduce": round(local_us, 2),
            "epoch_us_what": "wall per epoch of rank 0's shard: phase 1 + all-reduce + phase 2 (MAX over ranks) against the "
                             "same shard through the single-call epoch with no collective (rank 0)",
            "floats": n_floats, "loss_first": losses[0], "loss_last": losses[-1], "ranks_seen": ranks_seen,
            "data_unconverged": bad,
            "what": "per epoch: forward + loss + backward on the rank's shard, one all-reduce(SUM) of the flat fp32 "
                    "gradient + loss buffer, Adam + clamp + plateau schedule on every rank (physics_train.py:289-304)",
        }
        if check and world > 1:
            traj_all, _ = device_trajectories(torch, rr, ctl)
            rob1, _ = torch_rod(torch, dev, N, list(layers))
            solo = KnodeTrainer(rob1, traj_all, torch.as_tensor(ctl, device=dev).float().contiguous(), list(key_pts),
                                group=False)
            for _ in range(epochs):
                solo.step(sync_loss=False)
            torch.cuda.synchronize()
            l1 = np.asarray(solo.losses(), dtype=np.float64)
            ld = np.asarray(losses, dtype=np.float64)
            dev_rel = np.abs(ld - l1) / np.abs(l1)
            out["single_rank_check"] = {"loss_first": float(l1[0]), "loss_last": float(l1[-1]),
                                        "rel_dev_first": float(dev_rel[0]), "rel_dev_max": float(dev_rel.max()),
                                        "ok": bool(dev_rel[0] < 2e-5 and dev_rel.max() < 2e-4),
                                        "bar": "epoch 0 to 2e-5, every epoch to 2e-4 (fp32 sums in a different order, "
                                               "amplified by Adam over the epochs)"}
    if world > 1:
        dist.barrier()   # the other ranks wait for rank 0's single-rank repeat
    return out
