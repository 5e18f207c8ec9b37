#!/usr/bin/env python3
"""Dev tool (GPU box): Newton sweep statistics and step time of the shooting kernel
for a few tolerances.  python tools/iter_stats.py [B] [N] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
T = int(sys.argv[3]) if len(sys.argv) > 3 else 60
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native()
dev = "cuda:0"
ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 1235), device=dev)
import itertools
for (dt, tols), (ms, pred) in itertools.product(((torch.float64, (1e-8, 1e-10)), (torch.float32, (1e-5,))),
                                                ((0, 0), (0, 2), (1, 0), (1, 1), (1, 2))):
    h.set_option("ms_mode", ms); h.set_option("predictor", pred)
    ref_tip = None
    for tol in tols:
        c = ctl.to(dt).contiguous()
        st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0])
        G = torch.zeros((B, 6), dtype=dt, device=dev)
        its = torch.zeros((T, B), dtype=torch.int32, device=dev)
        sts = torch.zeros((T, B), dtype=torch.int32, device=dev)
        cs = [c[:, t].contiguous() for t in range(T)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(T):
            h.step(st[(t + 2) % 3 if t else 0], st[t % 3], st[(t + 1) % 3], G, cs[t], tol=tol, status=sts[t], iters=its[t],
                   prev2=st[(t + 1) % 3] if t >= 2 else None, predictor=min(pred, t))
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        tip = h.tip(st[T % 3]).cpu().numpy()
        if ref_tip is None and dt == torch.float64: pass
        it = its.cpu().numpy()
        hist = np.bincount(it[10:].ravel(), minlength=8)
        wave_max = it[10:].reshape(T - 10, -1, 8).max(axis=2).mean() if B % 8 == 0 else -1
        print(f"ms={ms} pred={pred} {str(dt):14s} tol={tol:7.0e} ms/step={el/T*1e3:7.3f} mean_it={it[10:].mean():.2f} wave_max_it={wave_max:.2f} "
              f"hist(it)={hist[:9].tolist()} bad={int((sts!=0).sum())} tip0={tip[0]}")
