#!/usr/bin/env python3
"""Dev tool (GPU box): |dG| per Newton iteration for a few rods, and what a
linear / quadratic extrapolation of G from previous steps would buy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
N = 100; T = 40; B = 4
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native(); dev = "cuda:0"; dt = torch.float64
ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 1235), device=dev).contiguous()
st = h.new_state(B, dt, n_slots=T + 1); h.init_straight(st[0])
G = torch.zeros((B, 6), dtype=dt, device=dev)
Gs = []
for t in range(T):
    h.step(st[t - 1 if t else 0], st[t], st[t + 1], G, ctl[:, t].contiguous(), tol=1e-12)
    Gs.append(G.clone())
Gs = torch.stack(Gs)  # [T,B,6]
def newton_trace(t, b, G0):
    prev = st[t - 1][b:b + 1].expand(7, -1, -1).contiguous(); cur = st[t][b:b + 1].expand(7, -1, -1).contiguous()
    nxt = h.new_state(7, dt); tens = ctl[b:b + 1, t].expand(7, -1).contiguous()
    Gk = G0.clone(); out = []
    for it in range(6):
        hs = 1e-7 * torch.clamp(Gk.abs(), min=1.0)
        GG = Gk[None].repeat(7, 1)
        for c in range(6): GG[c + 1, c] += hs[c]
        res = h.residual(GG.contiguous(), prev, cur, nxt, tens)
        J = ((res[1:] - res[0:1]) / hs[:, None]).T
        d = torch.linalg.solve(J, res[0])
        out.append(float(d.abs().max() / max(1.0, float(Gk.abs().max()))))
        Gk = Gk - d
    return out
for t in (20, 35):
    for b in (0, 2):
        g1, g2, g3 = Gs[t - 1, b], Gs[t - 2, b], Gs[t - 3, b]
        print(f"t={t} b={b} |G|={float(g1.abs().max()):.3f}")
        print("   warm   ", ["%.1e" % x for x in newton_trace(t, b, g1)])
        print("   linear ", ["%.1e" % x for x in newton_trace(t, b, 2 * g1 - g2)])
        print("   quadr  ", ["%.1e" % x for x in newton_trace(t, b, 3 * g1 - 3 * g2 + g3)])
