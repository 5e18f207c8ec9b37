#!/usr/bin/env python3
"""Dev experiment (GPU box): how well can the shooting unknowns of the next step be predicted from their
history?  Polynomial extrapolation (orders 0..7) vs. an adaptive linear predictor fitted per rod."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
from math import comb
N, B, T = 100, 1024, 400
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native()
ctl = orc.batch_sine_controls(B, T, r.del_t, 1235)
# the 32 fastest rods (shortest period): recover periods from the control signal's zero crossings is overkill - take
# the rods whose tension changes most per step
rough = np.abs(np.diff(ctl[:, :, 0], axis=1)).mean(axis=1)
sel = np.argsort(-rough)[:32]
c = torch.as_tensor(ctl[sel], device="cuda:0").contiguous()
st = h.new_state(len(sel), torch.float64, n_slots=T + 1); h.init_straight(st[0])
G = torch.zeros((len(sel), 6), dtype=torch.float64, device="cuda:0")
h.simulate(c, st, G); torch.cuda.synchronize()
S = st.cpu().numpy()  # [T+1, b, N, 28]  slots q w v u p h n m
starts = [0, 25, 50, 75]
rows = list(range(12, 25)) + list(range(0, 6))   # p h n m q w  (19)
X = np.concatenate([S[:, :, s, :][:, :, rows] for s in starts], axis=2)  # [T+1, b, 76]
X = X[60:]  # leave the start-up transient
def score(pred, x):  # scaled max norm per rod
    return np.max(np.abs(pred - x) / np.maximum(np.abs(x), 1.0), axis=-1)
Tn = X.shape[0]
res = {}
for p in range(8):
    w = np.array([(-1) ** k * comb(p + 1, k + 1) for k in range(p + 1)], dtype=float)
    e = []
    for t in range(10, Tn - 1):
        pred = sum(w[k] * X[t - k] for k in range(p + 1))
        e.append(score(pred, X[t + 1]))
    res[f"poly{p}"] = np.array(e)
for m, lam in ((3, 1e-12), (5, 1e-12), (7, 1e-12), (7, 1e-8), (7, 1e-6)):
    e = []
    b7 = np.zeros(m); pb = min(m - 1, 7)
    b7[: pb + 1] = [(-1) ** k * comb(pb + 1, k + 1) for k in range(pb + 1)]
    for t in range(10, Tn - 1):
        preds = np.empty_like(X[t])
        for b in range(X.shape[1]):
            wgt = 1.0 / np.maximum(np.abs(X[t, b]), 1.0)
            H = np.stack([X[t - 1 - k, b] for k in range(m)], axis=1) * wgt[:, None]   # [76, m]
            y = X[t, b] * wgt
            A = H.T @ H; sc = np.trace(A) / m
            cfs = np.linalg.solve(A + lam * sc * np.eye(m), H.T @ y + lam * sc * b7)
            preds[b] = np.stack([X[t - k, b] for k in range(m)], axis=1) @ cfs
        e.append(score(preds, X[t + 1]))
    res[f"lpc{m}_lam{lam:g}"] = np.array(e)
for k, v in res.items():
    print(f"{k:18s} median {np.median(v):.2e}  90% {np.quantile(v, 0.9):.2e}  max {v.max():.2e}   worst-rod median {np.median(v, axis=0).max():.2e}")
