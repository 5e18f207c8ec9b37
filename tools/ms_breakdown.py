#!/usr/bin/env python3
"""Dev tool (GPU box): time per Newton iteration of the multiple-shooting kernel, split into the part that
scales with N (sweep) and the part that does not (condensation / solve)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = "cuda:0"; dt = torch.float64
res = {}
for ms in (1, 0):
    for N in (101, 201, 401):
        r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
        h = r._native(); h.set_option("ms_mode", ms)
        ctl = torch.as_tensor(orc.batch_sine_controls(B, 4, r.del_t, 1235), device=dev).contiguous()
        st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0])
        c0 = ctl[:, 0].contiguous()
        for maxit in (1, 2, 3, 4):
            ts = []
            for rep in range(6):
                G = torch.zeros((B, 6), dtype=dt, device=dev)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10):
                    h.step(st[0], st[0], st[1], G, c0, tol=1e-30, maxit=maxit, predictor=0)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 10)
            res[(ms, N, maxit)] = min(ts) * 1e6
            print(f"ms={ms} N={N} maxit={maxit}: {min(ts)*1e6:8.1f} us")
for ms in (1, 0):
    for N in (101, 201, 401):
        per_it = (res[(ms, N, 4)] - res[(ms, N, 2)]) / 2
        print(f"ms={ms} N={N}: per Newton iteration {per_it:7.1f} us")
a = (res[(1, 101, 4)] - res[(1, 101, 2)]) / 2; b = (res[(1, 201, 4)] - res[(1, 201, 2)]) / 2
print(f"MS: sweep(100 segs/4)={b-a:.1f} us  algebra={2*a-b:.1f} us")
