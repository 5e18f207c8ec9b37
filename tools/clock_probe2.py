#!/usr/bin/env python3
"""Dev tool (GPU box): how the load that precedes a short (20-step) persistent launch sets its speed.
Steady-state trajectory (B=1024, N=100, fp64); before the timed launch the GPU runs `ramp` for 0.25 s."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench, krod_native as kn
from cosserat_ode import CosseratRod
from knode import setup_robot
B, N, K = 1024, 100, 20
dev = "cuda:0"; dt = torch.float64
def robot():
    r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms(); return r
r = robot(); h = r._native(); h.set_option("keep_predictor", 1)
r2 = robot(); h2 = r2._native()
ctl = torch.as_tensor(bench.rank_controls(B, 1, 0, 60 + K, r.del_t), device=dev).contiguous()
def ramp(kind, seconds):
    if kind == "none": return
    d = torch.float32 if kind == "f32" else torch.float64
    scheme = kn.KR_RK4 if kind == "f64rk4" else kn.KR_EULER
    c = ctl[:, :60].to(d).repeat(1, 4, 1).contiguous()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        s2 = h2.new_state(B, d, n_slots=3); h2.init_straight(s2[0]); g2 = torch.zeros((B, 6), dtype=d, device=dev)
        h2.simulate(c, s2, g2, ring=True, scheme=scheme)
        torch.cuda.synchronize()
for kind in ("none", "f32", "f64rk4", "f64", "none", "f64rk4"):
    res = []
    for rep in range(3):
        st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
        h.simulate(ctl[:, :60].contiguous(), st, G, ring=True)
        ck = ctl[:, 60:].contiguous(); pi = st[2].clone()
        ramp(kind, 0.25)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); h.simulate(ck, st, G, ring=True, prev_init=pi); e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / K)
    print(f"ramp {kind:7s}: us/step of the 20-step launch: " + " ".join(f"{x:.1f}" for x in res), flush=True)
