#!/usr/bin/env python3
"""GPU box, debugging aid: ONE kr_simulate_batch call with the MLP on and a given number of wavefronts per rod
(python tools/nn_w2_once.py B T W dtype)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
B, T, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dt = getattr(torch, sys.argv[4] if len(sys.argv) > 4 else "float64")
r = CosseratRod(use_fsolve=True); setup_robot(r, None); r.N = 100; r.compute_intermediate_terms()
mlp = orc.make_mlp([28, 64, 64, 25], "elu", seed=7)
model, params = [], []
for Wt, b, a in zip(mlp.weights, mlp.biases, mlp.acts):
    model.append("Linear"); params += [Wt, b]
    if a != orc.ACT_NONE: model.append("ELU(alpha=1.0)")
r.nn_model, r.param_ls, r.nn_path = model, params, "x"
h = r._native(); h.set_option("waves_per_rod", W)
ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 1235), device="cuda:0").to(dt).contiguous()
st = h.new_state(B, dt, n_slots=T + 1); h.init_straight(st[0]); Gs = torch.zeros((B, 6), dtype=dt, device="cuda:0")
status = torch.zeros((B, T), dtype=torch.int32, device="cuda:0")
h.simulate(ctl, st, Gs, use_nn=True, status=status); torch.cuda.synchronize()
print("ran W =", h.get_option("last_waves_per_rod"), "status", status.cpu().numpy().ravel()[:8], "tip", st[T, 0, -1, 12:15].cpu().numpy())
