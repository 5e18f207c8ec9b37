#!/usr/bin/env python3
"""Dev tool (GPU box): what a SHORT persistent launch costs beyond its steps.  One trajectory (B=1024, N=100, fp64) advanced
in back-to-back kr_simulate_batch calls of K steps each with the predictor handed over; prints the event time of every call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench
from cosserat_ode import CosseratRod
from knode import setup_robot
B, N = 1024, 100
dev = "cuda:0"; dt = torch.float64
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native()
for K in (20, 60, 200):
    chunks = 12
    ctl = torch.as_tensor(bench.rank_controls(B, 1, 0, 60 + K * chunks, r.del_t), device=dev).contiguous()
    h.set_option("keep_predictor", 0); h.set_option("keep_predictor", 1)
    st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
    status = torch.zeros((B, K), dtype=torch.int32, device=dev)
    h.simulate(ctl[:, :60].contiguous(), st, G, ring=True)   # 60 = 0 mod 3: the ring is back at slot 0
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(chunks + 1)]
    t0 = 60
    assert K % 3 != 0 or True
    evs[0].record()
    for c in range(chunks):
        # ring position: after t0 steps the current state sits in slot t0 % 3; rotate views so that the call starts at its slot 0
        cur = t0 % 3
        order = [cur, (cur + 1) % 3, (cur + 2) % 3]
        if cur != 0:
            st = st[order].contiguous()
        h.simulate(ctl[:, t0:t0 + K].contiguous(), st, G, ring=True, status=status, prev_init=st[2].clone())
        evs[c + 1].record()
        t0 += K
    torch.cuda.synchronize()
    ms = [evs[c].elapsed_time(evs[c + 1]) for c in range(chunks)]
    print(f"K={K}: per call ms", " ".join(f"{m:.3f}" for m in ms), f"| us/step median {np.median(ms)/K*1e3:.1f}  bad {int((status!=0).sum())}")
