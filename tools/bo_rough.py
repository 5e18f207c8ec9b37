#!/usr/bin/env python3
"""GPU box: base-only storing sweeps under rough conditions (random-walk tensions, fresh random tensions every step, large
network weights): unconverged counts and states with the option on / off, fp64 and fp32.   python tools/bo_rough.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import krod_native as kn, cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
dev = "cuda:0"
B, N, T = 512, 60, 100
rng = np.random.default_rng(11)
walk = 6.0 + np.cumsum(0.08 * rng.standard_normal((B, T, 4)), axis=1)
fresh = rng.uniform(2.0, 12.0, size=(B, T, 4))
jump = np.where((np.arange(T) // 10 % 2 == 0)[None, :, None], 4.0, 11.0) + 0.0 * walk
for wname, c in (("random walk", walk), ("fresh tensions every step", fresh), ("square wave 4 <-> 11", jump)):
    for seed, scale, act in ((7, 1.0, "elu"), (13, 2.5, "tanh"), (21, 3.0, "relu")):
        r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
        mlp = orc.make_mlp([28, 64, 64, 25], act, seed=seed)
        model, params = [], []
        for W_, b_, a_ in zip(mlp.weights, mlp.biases, mlp.acts):
            model.append("Linear"); params += [W_ * scale, b_]
            if a_ != orc.ACT_NONE: model.append({"elu": "ELU(alpha=1.0)", "tanh": "Tanh()", "relu": "ReLU()"}[act])
        r.nn_model, r.param_ls, r.nn_path = model, params, "x"
        h = r._native(); h.set_option("waves_per_rod", 1)
        for dts, dt in (("f64", torch.float64), ("f32", torch.float32)):
            ctl = torch.as_tensor(c, device=dev).to(dt).contiguous()
            res = []
            for on in (1, 0):
                h.set_option("nn_base_only_store", on)
                st = h.new_state(B, dt, n_slots=T + 1); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
                status = torch.zeros((B, T), dtype=torch.int32, device=dev)
                iters = torch.zeros((B, T), dtype=torch.int32, device=dev) if False else None
                h.simulate(ctl, st, G, ring=False, status=status, use_nn=True)
                torch.cuda.synchronize()
                s = status.cpu().numpy()
                res.append((st.double().cpu().numpy(), s))
            (a, sa), (b, sb) = res
            okrod = np.all(sa == 0, axis=1) & np.all(sb == 0, axis=1)
            fin = np.isfinite(a).all() and np.isfinite(b).all()
            if okrod.any():
                num = np.sqrt(((a[1:, okrod] - b[1:, okrod])[..., :25] ** 2).sum(axis=(2, 3)))
                den = np.sqrt((b[1:, okrod][..., :25] ** 2).sum(axis=(2, 3)))
                rel = float((num / den).max())
            else:
                rel = float("nan")
            print(f"{wname:26s} seed {seed:2d} x{scale} {act:5s} {dts}: statuses on {np.bincount(sa.ravel(), minlength=4)[:4]} off {np.bincount(sb.ravel(), minlength=4)[:4]}; "
                  f"rods converged throughout in both {int(okrod.sum())}/{B}: worst state rel L2 difference {rel:.2e}; finite {fin} (path {h.get_option('last_sim_path')})", flush=True)
