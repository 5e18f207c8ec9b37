#!/usr/bin/env python3
"""GPU box: which PERSISTENT kernel is fastest for short rods in batches that leave SIMDs idle - the overlapped one-wavefront
kernel (K2e) or several wavefronts per rod (K2d); steady state, 3-slot ring.   python tools/sel_timing.py [N B ...pairs]"""
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

def timed(h, B, T, dtype, warm=60):
    ctl = torch.as_tensor(orc.batch_sine_controls(B, warm + T, 0.005, 77), device=dev).to(dtype).contiguous()
    best = 1e9
    for _ in range(3):
        st = h.new_state(B, dtype, n_slots=3); h.init_straight(st[0])
        Gs = torch.zeros((B, 6), dtype=dtype, device=dev)
        status = torch.zeros((B, T), dtype=torch.int32, device=dev)
        h.simulate(ctl[:, :warm].contiguous(), st, Gs, ring=True)
        newest, older = st[warm % 3].clone(), st[(warm - 1) % 3].clone(); st[0].copy_(newest)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.simulate(ctl[:, warm:].contiguous(), st, Gs, ring=True, status=status, prev_init=older)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best / T, int((status != 0).sum())

args = [int(a) for a in sys.argv[1:]] or [24, 256, 32, 256, 40, 256, 56, 256, 64, 256, 80, 256, 100, 256, 101, 256, 32, 512, 40, 512, 64, 512, 100, 512, 100, 128, 100, 768]
for N, B in zip(args[0::2], args[1::2]):
    r = robot(N); h = r._native()
    h.set_option("keep_predictor", 1)
    line = f"N={N:4d} B={B:5d}:"
    for dt in (torch.float64,):
        for W in (0, 1, 2, 4):
            h.set_option("waves_per_rod", W)
            s, bad = timed(h, B, 120, dt)
            line += f"  W={W if W else 'auto'} ran (path {h.get_option('last_sim_path')}, W {h.get_option('last_waves_per_rod')}, overlap {h.get_option('last_overlap')}) {s*1e6:6.1f} us" + (f" bad {bad}" if bad else "")
    print(line, flush=True)
