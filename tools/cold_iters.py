#!/usr/bin/env python3
"""Dev tool (GPU box): sweeps per step over the first steps of a trajectory that starts from the straight rod
(bench workload, B=1024, N=100, fp64): where the cold-start figure of bench.py loses against the steady state."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import numpy as np, torch
import krod_native as kn
from cosserat_ode import CosseratRod
from knode import setup_robot
import bench
B, N, T = 1024, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 64
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native(); dt = torch.float64
ctl = torch.as_tensor(bench.rank_controls(B, 1, 0, T, r.del_t), device="cuda:0").contiguous()
h.set_option("persistent", 0)   # per-step launches report sweeps per (rod, step)
its = torch.zeros((B, T), dtype=torch.int32, device="cuda:0")
st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device="cuda:0")
kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(its)))
h.simulate(ctl, st, G, ring=True)
torch.cuda.synchronize()
kn.check(h.lib.kr_debug_buffer(h._h, None))
ii = its.cpu().numpy()
print("step: mean sweeps / max sweeps over the 1024 rods")
for t in range(T):
    print(f"{t:3d}: {ii[:, t].mean():.2f} / {ii[:, t].max()}", end="   " if (t + 1) % 6 else "\n")
print(f"\nsum over steps of the per-step max: {ii.max(axis=0).sum()}, of the mean: {ii.mean(axis=0).sum():.1f}; max over rods of the per-rod total: {ii.sum(axis=1).max()} (persistent launch time ~ this)")
