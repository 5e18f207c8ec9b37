#!/usr/bin/env python3
"""Dev tool (GPU box): audit of the residual estimate of ms_newton WITH THE MLP ON.  Needs
    make dbg DBGFLAGS="-DKR_MS_STAMPS -DKR_QUICK_AUDIT"
Prints, over all rods and steps, the worst ratio (Newton update norm of a storing sweep) / (amp x residual norm of that sweep)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import bench, krod_native as kn, cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
dev = "cuda:0"
for dts, dt in (("f64", torch.float64), ("f32", torch.float32)):
  for name, B, N, T, mk in (("bench workload", 1024, 100, 120, lambda B, T, d: bench.rank_controls(B, 1, 0, T, d)),
                            ("batch_sine 1237", 1024, 100, 120, lambda B, T, d: orc.batch_sine_controls(B, T, d, 1237)),
                            ("N=40 bench", 1024, 40, 120, lambda B, T, d: bench.rank_controls(B, 1, 0, T, d))):
    for seed, scale, act in ((7, 1.0, "elu"), (13, 1.5, "tanh"), (3, 0.3, "softplus")):
        r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
        mlp = orc.make_mlp([28, 64, 64, 25], act, seed=seed)
        model, params = [], []
        for W_, b_, a_ in zip(mlp.weights, mlp.biases, mlp.acts):
            model.append("Linear"); params += [W_ * scale, b_]
            if a_ != orc.ACT_NONE: model.append({"elu": "ELU(alpha=1.0)", "tanh": "Tanh()", "softplus": "Softplus(beta=1.0, threshold=20.0)"}[act])
        r.nn_model, r.param_ls, r.nn_path = model, params, "x"
        h = r._native(); h.set_option("ms_mode", 1); h.set_option("persistent", 1)
        ctl = torch.as_tensor(mk(B, T, r.del_t), device=dev).to(dt).contiguous()
        dbg = torch.zeros((B, 24), dtype=torch.int64, device=dev)
        kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(dbg)))
        st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
        status = torch.zeros((B, T), dtype=torch.int32, device=dev)
        h.simulate(ctl, st, G, ring=True, status=status, use_nn=True)
        torch.cuda.synchronize()
        q = dbg[:, 15].cpu().numpy().view(np.float64)
        its = dbg[:, 4].cpu().numpy().astype(np.float64)
        print(f"{dts} {name:18s} mlp seed {seed} x{scale} {act:8s}: worst ratio over rods {q.max():9.2f}  median {np.median(q):7.2f}  99.9% {np.quantile(q, 0.999):8.2f}  "
              f"sweeps/step {its.mean()/T:.2f}  unconverged {int((status != 0).sum())}", flush=True)
