#!/usr/bin/env python3
"""Dev tool (GPU box): long runs on non-smooth controls (fresh random tensions every step; random jumps), full batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from cosserat_ode import CosseratRod
from knode import setup_robot
B, N, T = 1024, 100, 400
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native()
rng = np.random.default_rng(0)
cases = {"random every step": 5.0 + 5.0 * rng.uniform(size=(B, T, 4))}
c = np.full((B, T, 4), 5.0)
for b in range(B):
    for t0 in rng.choice(np.arange(10, T - 10), size=6, replace=False):
        c[b, t0:, rng.integers(4)] += rng.uniform(-1.5, 1.5)
cases["random jumps"] = c
for name, ctl in cases.items():
    for dt in (torch.float64, torch.float32):
        ct = torch.as_tensor(ctl, device="cuda:0").to(dt).contiguous()
        st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device="cuda:0")
        status = torch.full((B, T), -1, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.simulate(ct, st, G, ring=True, status=status); torch.cuda.synchronize(); el = time.perf_counter() - t0
        s = status.cpu().numpy()
        print(f"{name:18s} {str(dt):14s}: status counts {np.bincount(s.ravel(), minlength=3)}  {el/T*1e6:.1f} us/step  finite={bool(torch.isfinite(st).all())}")
