#!/usr/bin/env python3
"""Dev tool (GPU box): sweeps per step with the MLP on (28-64-64-25 ELU, cfg3 inputs), single shooting (exact
forward-difference Jacobian of the 6 unknowns, sees the network's dependence on p) against multiple shooting (condensed
Jacobian without p columns, bf16 JVP for the network)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cosserat_oracle as orc, krod_native as kn
from cosserat_ode import CosseratRod
from knode import setup_robot
from gpu_helpers import inject
B, N, T = 128, 100, 60
dev = "cuda:0"; dt = torch.float64
for scale in (1.0, -1.0):  # -1: the network made blind to p (first-layer columns 0..2 zeroed), weights x1
    mlp = orc.make_mlp([28, 64, 64, 25], "elu", seed=7)
    if scale < 0:
        mlp.weights[0] = mlp.weights[0].copy(); mlp.weights[0][:, 0:3] = 0.0
    for mode in (1,):
        r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms(); inject(r, mlp)
        h = r._native(); h.set_option("ms_mode", mode); h.set_option("persistent", 0)
        from cosserat_ode import mlp_from_layer_strings
        h.set_mlp(*mlp_from_layer_strings(r.nn_model, r.param_ls))
        ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 1235), device=dev).contiguous()
        its = torch.zeros((B, T), dtype=torch.int32, device=dev)
        kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(its)))
        st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
        status = torch.zeros((B, T), dtype=torch.int32, device=dev)
        h.simulate(ctl, st, G, ring=True, use_nn=True, status=status)
        torch.cuda.synchronize()
        ii = its.cpu().numpy()
        print(f"{'p-blind network' if scale < 0 else 'network as is  '}: {'multiple' if mode else 'single  '} shooting (path {h.get_option('last_sim_path')}): sweeps per step, steps 30..: mean {ii[:, 30:].mean():.2f} "
              f"worst rod {ii[:, 30:].mean(axis=1).max():.2f} hist {np.bincount(ii[:, 30:].ravel(), minlength=8)[:8]} unconverged {int((status != 0).sum())}", flush=True)
