#!/usr/bin/env python3
"""Dev tool (GPU box): what should precede the timed 20-step launch of bench.py?  Event time per step of the launch
after: a heavy ramp (1 s of fp64 RK4 launches), then a pause / a light phase of varying length."""
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
c2 = ctl[:, :60].repeat(1, 2, 1)[:, :100].contiguous()
s2 = h2.new_state(B, dt, n_slots=3); g2 = torch.zeros((B, 6), dtype=dt, device=dev)
def heavy(seconds, sync_each):
    n = max(1, int(seconds / 0.012))
    for _ in range(n):
        h2.init_straight(s2[0]); g2.zero_()
        h2.simulate(c2, s2, g2, ring=True, scheme=kn.KR_RK4)
        if sync_each: torch.cuda.synchronize()
    torch.cuda.synchronize()
def light(n):  # n short launches of the same solver (Euler, 5 steps) with host gaps
    for _ in range(n):
        h2.init_straight(s2[0]); g2.zero_()
        h2.simulate(c2[:, :5].contiguous(), s2, g2, ring=True)
        torch.cuda.synchronize()
cases = [("heavy 1.0 s async", lambda: heavy(1.0, False)), ("heavy 1.0 s, sync each", lambda: heavy(1.0, True)),
         ("heavy 1.0 s + sleep 2 ms", lambda: (heavy(1.0, False), time.sleep(0.002))),
         ("heavy 1.0 s + sleep 20 ms", lambda: (heavy(1.0, False), time.sleep(0.02))),
         ("heavy 1.0 s + sleep 200 ms", lambda: (heavy(1.0, False), time.sleep(0.2))),
         ("heavy 1.0 s + 20 light launches", lambda: (heavy(1.0, False), light(20))),
         ("heavy 0.1 s async", lambda: heavy(0.1, False)), ("nothing", lambda: None), ("heavy 3 s async", lambda: heavy(3.0, False))]
for name, fn in cases:
    res = []
    for rep in range(3):
        st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
        h.simulate(ctl[:, :60].contiguous(), st, G, ring=True)
        ck = ctl[:, 60:].contiguous(); pi = st[2].clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        e0.record(); h.simulate(ck, st, G, ring=True, prev_init=pi); e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / K)
    print(f"{name:34s}: us/step of the 20-step launch: " + " ".join(f"{x:.1f}" for x in res), flush=True)
