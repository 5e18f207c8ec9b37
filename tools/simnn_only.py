#!/usr/bin/env python3
"""Dev tool (GPU box): forward simulation with the MLP inside the sweeps, alone, for profiling."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
dev = "cuda:0"
dt = torch.float64 if (len(sys.argv) < 2 or sys.argv[1] == "f64") else torch.float32
B, N, TS = 1024, 100, int(sys.argv[2]) if len(sys.argv) > 2 else 12
rr = CosseratRod(use_fsolve=True); setup_robot(rr); rr.N = N; rr.compute_intermediate_terms()
mlp = orc.make_mlp([28, 64, 64, 25], "elu", seed=7)
model, params = [], []
for W, b, a in zip(mlp.weights, mlp.biases, mlp.acts):
    model.append("Linear"); params += [W, b]
    if a != orc.ACT_NONE: model.append("ELU(alpha=1.0)")
rr.nn_model, rr.param_ls, rr.nn_path = model, params, "x"
h = rr._native()
ctl = torch.as_tensor(orc.batch_sine_controls(B, TS, rr.del_t, 1235), device=dev).to(dt).contiguous()
st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); Gs = torch.zeros((B, 6), dtype=dt, device=dev)
status = torch.zeros((B, TS), dtype=torch.int32, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
h.simulate(ctl, st, Gs, ring=True, use_nn=True, status=status)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / TS
print(f"{dt}: {el*1e3:.3f} ms/step bad={int((status!=0).sum())}")
