#!/usr/bin/env python3
"""Dev tool (GPU box): the step kernels against each other on long, rough inputs.  Every kernel solves the same
discrete equations; their results may differ by the stopping tolerance only.  Checks, per case, that every step of
every rod converges and that the tip trajectories agree with the single-shooting kernel (which has neither the
condensation, nor the chord / residual tests, nor the predictor of the multiple-shooting ones).
   python tools/soak_kernels.py            (about a minute)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
dev = "cuda:0"

def controls(kind, B, T, rng, del_t):
    if kind == "random every step":
        return 5.0 + 5.0 * rng.uniform(size=(B, T, 4))
    if kind == "random jumps":
        c = np.full((B, T, 4), 5.0)
        for b in range(B):
            for t0 in rng.choice(np.arange(5, T - 5), size=5, replace=False):
                c[b, t0:, rng.integers(4)] += rng.uniform(-1.5, 1.5)
        return c
    if kind == "random walk":
        return 6.0 + np.cumsum(0.08 * rng.standard_normal((B, T, 4)), axis=1)
    return orc.batch_sine_controls(B, T, del_t, int(rng.integers(1 << 30)))

def run(h, ctl, dt, opts):
    for k, v in opts.items(): h.set_option(k, v)
    B, T = ctl.shape[:2]
    ct = torch.as_tensor(ctl, device=dev).to(dt).contiguous()
    st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
    status = torch.full((B, T), -1, dtype=torch.int32, device=dev)
    tip = torch.empty((B, T, 3), dtype=dt, device=dev)
    h.simulate(ct, st, G, ring=True, status=status, tip=tip)
    torch.cuda.synchronize()
    return tip.double().cpu().numpy(), int((status != 0).sum()), (h.get_option("last_sim_path"), h.get_option("last_waves_per_rod"))

rng = np.random.default_rng(2026)
worst = 0.0
for N, B, T in ((100, 512, 240), (40, 512, 240), (130, 192, 120), (257, 128, 80), (400, 96, 60)):
    for mod in (None, "dampstiff"):
        r = CosseratRod(use_fsolve=True); setup_robot(r, mod); r.N = N; r.compute_intermediate_terms()
        h = r._native()
        for kind in ("random every step", "random jumps", "random walk", "sine"):
            ctl = controls(kind, B, T, rng, r.del_t)
            for dt in (torch.float64, torch.float32):
                ref, bad0, p0 = run(h, ctl, dt, {"ms_mode": 0, "persistent": 0, "waves_per_rod": 1})
                variants = [("multi, per step", {"ms_mode": 1, "persistent": 0, "waves_per_rod": 1})]
                if N <= 128: variants.append(("multi, persistent", {"ms_mode": 1, "persistent": 1, "waves_per_rod": 1}))
                if N - 1 >= 14: variants.append(("2 wavefronts", {"ms_mode": 1, "persistent": 0, "waves_per_rod": 2}))
                if N - 1 >= 26: variants.append(("4 wavefronts", {"ms_mode": 1, "persistent": 0, "waves_per_rod": 4}))
                if N - 1 >= 14 and N <= 257: variants.append(("2 wavefronts, persistent", {"ms_mode": 1, "persistent": 1, "waves_per_rod": 2}))
                if N - 1 >= 26 and N <= 257: variants.append(("4 wavefronts, persistent", {"ms_mode": 1, "persistent": 1, "waves_per_rod": 4}))
                line = []
                for name, opts in variants:
                    tip, bad, path = run(h, ctl, dt, opts)
                    scale = np.abs(ref).max()
                    err = np.abs(tip - ref).max() / scale
                    worst = max(worst, err if dt == torch.float64 else 0.0)
                    tol = 2e-7 if dt == torch.float64 else 2e-4
                    flag = "" if (bad == 0 and err < tol) else "  <-- CHECK"
                    line.append(f"{name} {path}: bad {bad} err {err:.1e}{flag}")
                print(f"N={N:3d} mod={str(mod):9s} {kind:18s} {str(dt)[6:]:8s} single bad {bad0} | " + " | ".join(line), flush=True)
        h.set_option("ms_mode", -1); h.set_option("persistent", 1); h.set_option("waves_per_rod", 0)
print(f"worst fp64 deviation from the single-shooting kernel, relative to the tip range: {worst:.2e}")
