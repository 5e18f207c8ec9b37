#!/usr/bin/env python3
"""Dev tool (GPU box): per-phase s_memtime ticks of the several-wavefront step kernel (kr_msw_impl.hpp).
Needs the diagnostic library: KR_LIB_PATH=knode-cosserat_amd/lib/dbg/libknode_rod.so
    python tools/msw_stamps.py N B W"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc, krod_native as kn
from cosserat_ode import CosseratRod
from knode import setup_robot
N, B, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (400, 512, 2)
T, warm = 40, 40
dev = "cuda:0"; dt = torch.float64
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native(); h.set_option("persistent", 0); h.set_option("keep_predictor", 1); h.set_option("waves_per_rod", W)
dbg = torch.zeros((16, B, T), dtype=torch.int32, device=dev)
ctl = torch.as_tensor(orc.batch_sine_controls(B, warm + T, r.del_t, 77), device=dev).contiguous()
st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
h.simulate(ctl[:, :warm].contiguous(), st, G, ring=True)
kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(dbg)))
torch.cuda.synchronize(); t0 = time.perf_counter()
h.simulate(ctl[:, warm:].contiguous(), st, G, ring=True, prev_init=st[2])
torch.cuda.synchronize(); el = time.perf_counter() - t0
d = dbg.cpu().numpy().astype(np.float64)
dnv = dbg[10:15].cpu().numpy().view(np.float32)
qa = dbg[15].cpu().numpy().view(np.float32)
names = ["sweeps", "total", "prologue", "sweep", "algebra", "pred update+save", "alg: hand-over+local chain", "alg: barrier wait",
         "alg: combine", "alg: solve+update+norm"]
print(f"N={N} B={B} W={W} (ran {h.get_option('last_waves_per_rod')}): wall {el/T*1e6:.1f} us/step (with the debug stores)")
ii = d[0].astype(int)
print(f"  sweeps per rod-step hist {np.bincount(ii.ravel(), minlength=6)}; per-step max over rods: mean {ii.max(axis=0).mean():.2f}; "
      f"total ticks: mean over steps of max over rods {d[1].max(axis=0).mean():.0f} -> {d[1].max(axis=0).mean()/(el/T*1e6):.0f} ticks/us if the kernel filled the wall time")
for k, n in enumerate(names):
    print(f"  {n:28s} mean {d[k].mean():9.1f}  max {d[k].max():9.1f}" + ("" if k == 0 else "  ticks (10 ns)"))
for k in (2, 3, 4):
    m = ii == k
    if m.any():
        print(f"  rod-steps with {k} sweeps: median update norms " + " ".join(f"{np.median(dnv[q][m]):.2e}" for q in range(4)) + f"  kappa {np.median(dnv[4][m]):.2e}")
if qa.max() > 0:
    print(f"  residual-test audit (-DKR_QUICK_AUDIT): (update) / (estimate) worst {qa.max():.1f}, median of the audited steps {np.median(qa[qa > 0]):.1f}, audited {int((qa > 0).sum())} of {qa.size}")
