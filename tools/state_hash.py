#!/usr/bin/env python3
"""GPU box: SHA-256 of the full state trajectory and the tips of a fixed headline-kernel run (B = 256, N = 100, T = 60, fp64 and
fp32, sine tensions from the straight rod) - to check that a change that claims to be bit-exact is.
    KR_LIB_PATH=<lib> python tools/state_hash.py"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench_legs as bl
dev = "cuda:0"
for dt in (torch.float64, torch.float32):
    B, N, T = 1024, 100, 60
    r = bl.make_robot(N, 0); h = r._native()
    ctl = torch.as_tensor(bl.sine_controls(B, T, r.del_t, 99), device=dev).to(dt).contiguous()
    st = h.new_state(B, dt, n_slots=T + 1); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
    tip = torch.empty((B, T, 3), dtype=dt, device=dev); status = torch.zeros((B, T), dtype=torch.int32, device=dev)
    h.simulate(ctl, st, G, tip=tip, status=status)
    torch.cuda.synchronize()
    a = st[..., :25].contiguous().cpu().numpy()
    print(str(dt), "overlap", h.get_option("last_overlap"), "states", hashlib.sha256(a.tobytes()).hexdigest()[:16], "tips", hashlib.sha256(tip.cpu().numpy().tobytes()).hexdigest()[:16],
          "unconverged", int((status != 0).sum()))
