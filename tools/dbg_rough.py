#!/usr/bin/env python3
"""Dev tool (GPU box): convergence of the step solvers on non-smooth controls."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from cosserat_ode import CosseratRod
from knode import setup_robot
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native()
B, T = 16, 90
rng = np.random.default_rng(5)
ctl = 5.0 + 5.0 * rng.uniform(size=(B, T, 4))
dev = "cuda:0"
ctl_t = torch.as_tensor(ctl, device=dev).contiguous()
ref = None
for ms, pers, pred in ((0, 0, 0), (0, 0, 2), (1, 0, 0), (1, 0, 2), (1, 1, 0), (1, 1, 2), (1, 1, 7), (1, 1, 8)):
    h.set_option("ms_mode", ms); h.set_option("persistent", pers); h.set_option("predictor", pred)
    st = h.new_state(B, torch.float64, n_slots=T + 1); h.init_straight(st[0])
    G = torch.zeros((B, 6), dtype=torch.float64, device=dev)
    status = torch.full((B, T), -1, dtype=torch.int32, device=dev)
    h.simulate(ctl_t, st, G, status=status); torch.cuda.synchronize()
    s = status.cpu().numpy()
    out = st[..., :25].cpu().numpy()
    if ref is None: ref = out
    d = np.abs(out - ref).max()
    print(f"ms={ms} persistent={pers} pred={pred}: status counts {np.bincount(s.ravel(), minlength=3)}  first bad step {np.argwhere(s != 0)[:3].tolist()}  max|diff vs first| {d:.2e}")
