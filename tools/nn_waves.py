#!/usr/bin/env python3
"""GPU box: kr_simulate_batch with the MLP inside every sweep (cfg3: N = 100, 28 -> 64 -> 64 -> 25 ELU, random weights)
with 1, 2 and 4 wavefronts per rod - time per step, unconverged steps, and the trajectories of 2 / 4 wavefronts
against the one-wavefront kernel's (both solve the same BVP to the same tolerance; they differ by what the tolerance
leaves).
    python tools/nn_waves.py [B=1024] [T=60] [layers=64,64] [act=elu]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 60
layers = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "64,64").split(",")]
act = sys.argv[4] if len(sys.argv) > 4 else "elu"
N = int(os.environ.get("KR_NW_N", "100"))
TOL = {"tol": float(os.environ["KR_NW_TOL"])} if "KR_NW_TOL" in os.environ else {}
ACTN = {"elu": "ELU(alpha=1.0)", "tanh": "Tanh()", "relu": "ReLU()", "softplus": "Softplus(beta=1.0, threshold=20.0)"}


def robot():
    r = CosseratRod(use_fsolve=True); setup_robot(r, None); r.N = N; r.compute_intermediate_terms()
    mlp = orc.make_mlp([28] + layers + [25], act, seed=7)
    model, params = [], []
    for W, b, a in zip(mlp.weights, mlp.biases, mlp.acts):
        model.append("Linear"); params += [W, b]
        if a != orc.ACT_NONE: model.append(ACTN[act])
    r.nn_model, r.param_ls, r.nn_path = model, params, "x"
    return r


r = robot()
h = r._native()
for dt in ((torch.float64, torch.float32) if "KR_NW_DT" not in os.environ else (getattr(torch, os.environ["KR_NW_DT"]),)):
    ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 1235), device=dev).to(dt).contiguous()
    ref = None
    for W in (1, 2, 4):
        h.set_option("waves_per_rod", W)
        best, out = 1e9, None
        for rep in range(2):
            st = h.new_state(B, dt, n_slots=T + 1); h.init_straight(st[0]); Gs = torch.zeros((B, 6), dtype=dt, device=dev)
            status = torch.zeros((B, T), dtype=torch.int32, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h.simulate(ctl, st, Gs, use_nn=True, status=status, **TOL)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / T)
        used = h.get_option("last_waves_per_rod"); path = h.get_option("last_sim_path")
        x = st[:, :, :, :25].cpu().numpy()
        if ref is None: ref = x
        d = float(np.max(np.abs(x - ref))); fin = bool(np.isfinite(x).all())
        print(f"{str(dt):14s} asked W={W} ran W={used} path {path}: {best*1e3:7.3f} ms/step -> {B/best/1e3:8.1f} k rod-steps/s, "
              f"unconverged {int((status != 0).sum())}, max |x - x(W=1)| = {d:.2e}, finite {fin}", flush=True)
    h.set_option("waves_per_rod", 0)
