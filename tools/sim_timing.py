#!/usr/bin/env python3
"""Dev tool (GPU box): per-step time when the time loop runs inside kr_simulate_batch (one C call)
vs one Python call per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
T = 120
dev = "cuda:0"
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native()
for dt in (torch.float64, torch.float32):
    ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 1235), device=dev).to(dt).contiguous()
    for ms, pers in ((0, 0), (1, 0), (1, 1)):
        for pred in (0, 7, 8):
            h.set_option("ms_mode", ms); h.set_option("predictor", pred); h.set_option("persistent", pers)
            best = 1e9
            for rep in range(3):
                st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0])
                G = torch.zeros((B, 6), dtype=dt, device=dev)
                status = torch.zeros((B, T), dtype=torch.int32, device=dev)
                h.simulate(ctl[:, :21].contiguous(), st, G, ring=True)  # warm-up / leave transient (21 = 0 mod 3)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                h.simulate(ctl, st, G, ring=True, status=status, prev_init=st[2])
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            print(f"{str(dt):14s} ms={ms} persistent={pers} pred={pred}: {best/T*1e6:7.1f} us/step  -> {B*T/best/1e6:6.2f} M rod-steps/s  bad={int((status!=0).sum())}")
