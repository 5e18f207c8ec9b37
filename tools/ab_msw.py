#!/usr/bin/env python3
"""GPU box: the several-wavefront configurations (cfg2: B=256 N=100; cfg5: B=512 N=400; B=512 N=100) with the library
KR_LIB_PATH points to - run once per library for an A/B of a change.   python tools/ab_msw.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
dev = "cuda:0"
def robot(N):
    r = CosseratRod(use_fsolve=True); setup_robot(r, None); r.N = N; r.compute_intermediate_terms(); return r
def timed(r, B, T, dt, warm=40, reps=4):
    h = r._native()
    ctl = torch.as_tensor(orc.batch_sine_controls(B, warm + T, r.del_t, 1234), device=dev).to(dt).contiguous()
    best = 1e9
    for _ in range(reps):
        st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
        status = torch.zeros((B, T), dtype=torch.int32, device=dev)
        h.simulate(ctl[:, :warm].contiguous(), st, G, ring=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.simulate(ctl[:, warm:].contiguous(), st, G, ring=True, status=status, prev_init=st[2])
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best / T * 1e6, h.get_option("last_sim_path"), h.get_option("last_waves_per_rod"), int((status != 0).sum())
print("library:", os.environ.get("KR_LIB_PATH", "(default)"))
for N, B, T in ((100, 256, 200), (100, 512, 200), (400, 512, 60), (400, 256, 60)):
    r = robot(N)
    for dt in (torch.float64, torch.float32):
        us, path, w, bad = timed(r, B, T, dt)
        print(f"  N={N} B={B} {str(dt):14s}: {us:7.1f} us/step (path {path}, {w} wavefronts per rod, unconverged {bad})", flush=True)
