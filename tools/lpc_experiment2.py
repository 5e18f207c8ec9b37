#!/usr/bin/env python3
"""Dev experiment (GPU box): floor of linear prediction of the shooting unknowns (more taps, longer fit windows)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
N, B, T = 100, 1024, 700
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native()
ctl = orc.batch_sine_controls(B, T, r.del_t, 1235)
rough = np.abs(np.diff(ctl[:, :, 0], axis=1)).mean(axis=1)
order = np.argsort(-rough)
sel = np.concatenate([order[:8], order[500:504], order[-4:]])
c = torch.as_tensor(ctl[sel], device="cuda:0").contiguous()
st = h.new_state(len(sel), torch.float64, n_slots=T + 1); h.init_straight(st[0])
G = torch.zeros((len(sel), 6), dtype=torch.float64, device="cuda:0")
h.simulate(c, st, G, tol=1e-12); torch.cuda.synchronize()
S = st.cpu().numpy()
starts = [0, 25, 50, 75]
rows = list(range(12, 25)) + list(range(0, 6))
X = np.concatenate([S[:, :, s, :][:, :, rows] for s in starts], axis=2)  # [T+1, b, 76]
def score(pred, x): return np.max(np.abs(pred - x) / np.maximum(np.abs(x), 1.0), axis=-1)
for lo, hi in ((80, 200), (400, 690)):
    print(f"--- steps {lo}..{hi}")
    for m in (3, 5, 7, 9):
        for R in (1, 4, 12):
            errs = []
            for t in range(lo, hi, 3):
                e = np.zeros(X.shape[1])
                for b in range(X.shape[1]):
                    wgt = 1.0 / np.maximum(np.abs(X[t, b]), 1.0)
                    Hs, ys = [], []
                    for rr in range(R):
                        Hs.append(np.stack([X[t - rr - 1 - k, b] for k in range(m)], axis=1) * wgt[:, None])
                        ys.append(X[t - rr, b] * wgt)
                    Hm = np.concatenate(Hs); ym = np.concatenate(ys)
                    cfs = np.linalg.lstsq(Hm, ym, rcond=1e-13)[0]
                    pred = np.stack([X[t - k, b] for k in range(m)], axis=1) @ cfs
                    e[b] = score(pred, X[t + 1, b])
                errs.append(e)
            errs = np.array(errs)
            print(f"m={m} rows={R:2d}: fast rods median {np.median(errs[:, :8]):.1e} max {errs[:, :8].max():.1e} | mid {np.median(errs[:, 8:12]):.1e} max {errs[:, 8:12].max():.1e} | slow {np.median(errs[:, 12:]):.1e} max {errs[:, 12:].max():.1e}")
