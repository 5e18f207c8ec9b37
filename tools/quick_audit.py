#!/usr/bin/env python3
"""Dev tool (GPU box): audit of the residual test of ms_newton.  Needs the library built with
    make dbg DBGFLAGS="-DKR_MS_STAMPS -DKR_QUICK_AUDIT"
(the test is then evaluated but never taken; the chord update that follows is compared with its estimate).
Prints, over all rods and steps, the worst ratio (chord update norm) / (amp x residual norm): the safety factor of 64
in kr_ms_impl.hpp has to cover it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import bench, krod_native as kn, cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
dev = "cuda:0"; dt = torch.float64
for name, B, N, T, mk in (("bench workload", 1024, 100, 300, lambda B, T, d: bench.rank_controls(B, 1, 0, T, d)),
                          ("batch_sine 1237", 1024, 100, 300, lambda B, T, d: orc.batch_sine_controls(B, T, d, 1237)),
                          ("N=40 bench", 1024, 40, 300, lambda B, T, d: bench.rank_controls(B, 1, 0, T, d)),
                          ("random walk tensions", 512, 100, 200, None)):
    for mod in (None, "dampstiff"):
        r = CosseratRod(use_fsolve=True); setup_robot(r, mod); r.N = N; r.compute_intermediate_terms()
        h = r._native()
        if mk is None:
            rng = np.random.default_rng(5)
            c = 6.0 + np.cumsum(0.05 * rng.standard_normal((B, T, 4)), axis=1)
        else:
            c = mk(B, T, r.del_t)
        ctl = torch.as_tensor(c, device=dev).contiguous()
        dbg = torch.zeros((B, 24), dtype=torch.int64, device=dev)
        kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(dbg)))
        st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
        status = torch.zeros((B, T), dtype=torch.int32, device=dev)
        h.simulate(ctl, st, G, ring=True, status=status)
        torch.cuda.synchronize()
        assert h.get_option("last_sim_path") == 2
        q = dbg[:, 15].cpu().numpy().view(np.float64)
        print(f"{name:22s} mod={str(mod):9s}: worst ratio over rods {q.max():8.2f}  median {np.median(q):6.2f}  99.9% {np.quantile(q, 0.999):7.2f}  unconverged {int((status != 0).sum())}", flush=True)
