#!/usr/bin/env python3
"""Dev tool (GPU box): latency of the drop-in surface for ONE rod (BASELINE cfg1: N=20, 200 steps) - knode.simulate wall
time per step on a warm process, where it goes (cProfile), and the same with the residual MLP injected."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot, simulate
from gpu_helpers import inject
ctl = [[6.0, 5.0, 5.0, 6.0]] * 200
for nn in (False, True):
    r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = 20; r.compute_intermediate_terms()
    if nn: inject(r, orc.make_mlp([28, 64, 25], "elu", seed=1))
    simulate(r, ctl)  # first call: handle creation, module load
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); simulate(r, ctl); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{'MLP on ' if nn else 'MLP off'}: knode.simulate 200 steps: best {min(ts)*1e3:.2f} ms = {min(ts)/199*1e3:.4f} ms/step (runs {[round(t*1e3,2) for t in ts]})")
    if not nn:
        pr = cProfile.Profile(); pr.enable(); simulate(r, ctl); torch.cuda.synchronize(); pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:2500])
