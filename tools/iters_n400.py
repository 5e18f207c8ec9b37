#!/usr/bin/env python3
"""Dev tool (GPU box): sweeps per rod and step on the one-launch-per-step path (N=400, B=512)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc, krod_native as kn
from cosserat_ode import CosseratRod
from knode import setup_robot
B, N, W, K = 512, 400, 30, 60
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native()
dt = torch.float64
ctl = torch.as_tensor(orc.batch_sine_controls(B, W + K, r.del_t, 1237), device="cuda:0").contiguous()
st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device="cuda:0")
h.set_option("keep_predictor", 1)
h.simulate(ctl[:, :W].contiguous(), st, G, ring=True)
its = torch.zeros((B, K), dtype=torch.int32, device="cuda:0")
kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(its)))
torch.cuda.synchronize(); t0 = time.perf_counter()
h.simulate(ctl[:, W:].contiguous(), st, G, ring=True, prev_init=st[2])
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / K
kn.check(h.lib.kr_debug_buffer(h._h, None))
ii = its.cpu().numpy()
print(f"{el*1e6:.1f} us/step; sweeps per rod-step: mean {ii.mean():.2f}; per-step max over rods: mean {ii.max(axis=0).mean():.2f}; worst rod mean {ii.mean(axis=1).max():.2f}; hist {np.bincount(ii.ravel(), minlength=6)}")
