import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
B, N, T = 1024, 100, 40
dev = "cuda:0"; dt = torch.float64
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native(); h.set_option("ms_mode", 1)
ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 1235), device=dev).contiguous()
st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0])
G = torch.zeros((B, 6), dtype=dt, device=dev)
h.simulate(ctl, st, G, ring=True)
torch.cuda.synchronize()
# fixed iteration counts (never converge): maxit = 1..4
c0 = ctl[:, 0].contiguous()
for maxit in (1, 2, 3, 4, 1, 2, 3, 4):
    G.zero_()
    h.step(st[0], st[0], st[1], G, c0, tol=1e-30, maxit=maxit, predictor=0)
torch.cuda.synchronize()
