#!/usr/bin/env python3
"""GPU box: BASELINE cfg5 alone (B=512, N=400, fp64, sine tensions), for rocprofv3: 30 warm-up + 60 steps, one launch per
step on two wavefronts per rod (kr_msw_impl.hpp).    python tools/cfg5_only.py [f64|f32]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.float64
B, N, W, T = 512, 400, 30, 60
r = CosseratRod(use_fsolve=True); setup_robot(r, None); r.N = N; r.compute_intermediate_terms()
h = r._native(); h.set_option("keep_predictor", 1)
ctl = torch.as_tensor(orc.batch_sine_controls(B, W + T, r.del_t, 1237), device="cuda:0").to(dt).contiguous()
st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device="cuda:0")
status = torch.zeros((B, T), dtype=torch.int32, device="cuda:0")
h.simulate(ctl[:, :W].contiguous(), st, G, ring=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
h.simulate(ctl[:, W:].contiguous(), st, G, ring=True, status=status, prev_init=st[2])
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / T
print(f"cfg5 {dt}: {el*1e6:.1f} us/step, {B/el/1e6:.2f} M rod-steps/s, path {h.get_option('last_sim_path')}, "
      f"{h.get_option('last_waves_per_rod')} wavefronts per rod, unconverged {int((status != 0).sum())}; "
      f"algorithmic bytes per step (75 N + 16) s B = {(75*N+16)*(8 if dt == torch.float64 else 4)*B/1e6:.1f} MB")
