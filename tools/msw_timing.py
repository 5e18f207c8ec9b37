#!/usr/bin/env python3
"""GPU box: time per step of the one-launch-per-step multiple-shooting kernel with 1, 2 and 4 wavefronts per rod
(kr_msw_impl.hpp) for long rods / small batches.   python tools/msw_timing.py [N B ...pairs]"""
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

def timed(h, B, T, dtype, warm=40):
    ctl = torch.as_tensor(orc.batch_sine_controls(B, warm + T, 0.005, 77), device=dev).to(dtype).contiguous()
    best = 1e9
    for _ in range(3):
        st = h.new_state(B, dtype, n_slots=3); h.init_straight(st[0])
        Gs = torch.zeros((B, 6), dtype=dtype, device=dev)
        status = torch.zeros((B, T), dtype=torch.int32, device=dev)
        h.simulate(ctl[:, :warm].contiguous(), st, Gs, ring=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.simulate(ctl[:, warm:].contiguous(), st, Gs, ring=True, status=status, prev_init=st[2])
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best / T, int((status != 0).sum()), st[0].double().cpu().numpy()

args = [int(a) for a in sys.argv[1:]] or [400, 512, 400, 256, 400, 64, 100, 256, 100, 64, 200, 256]
for N, B in zip(args[0::2], args[1::2]):
    r = robot(N); h = r._native()
    h.set_option("persistent", 0); h.set_option("keep_predictor", 1)
    for dt in (torch.float64, torch.float32):
        ref = None
        for persist, W in ((1, 1), (0, 1), (0, 2), (0, 4), (1, 2), (1, 4)):  # persistent: all steps in one launch
            h.set_option("persistent", persist)
            h.set_option("waves_per_rod", W)
            s, bad, last = timed(h, B, 60, dt)
            got = f"{'persistent' if h.get_option('last_sim_path') == 2 else 'per step  '} {h.get_option('last_waves_per_rod')}"
            ref = last if ref is None else ref
            print(f"N={N:4d} B={B:5d} {str(dt):14s} {('persistent' if persist else 'per step  ')} W={W} (ran {got})  {s*1e6:8.1f} us/step  {B/s/1e6:6.2f} M rod-steps/s  "
                  f"unconverged {bad}  max|d| vs W=1 {np.abs(last-ref).max():.2e}", flush=True)
