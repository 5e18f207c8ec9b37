#!/usr/bin/env python3
"""Dev tool (GPU box): NN-on forward simulation, time per step after the start-up transient."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
dev = "cuda:0"
B, N = 1024, 100
rr = CosseratRod(use_fsolve=True); setup_robot(rr); rr.N = N; rr.compute_intermediate_terms()
mlp = orc.make_mlp([28, 64, 64, 25], "elu", seed=7)
model, params = [], []
for W, b, a in zip(mlp.weights, mlp.biases, mlp.acts):
    model.append("Linear"); params += [W, b]
    if a != orc.ACT_NONE: model.append("ELU(alpha=1.0)")
rr.nn_model, rr.param_ls, rr.nn_path = model, params, "x"
h = rr._native()
for dt in (torch.float64, torch.float32):
    for pred in (0, 2, 8):
        h.set_option("predictor", pred)
        h.set_option("keep_predictor", 0); h.set_option("keep_predictor", 1)
        W, K = 40, 40
        ctl = torch.as_tensor(orc.batch_sine_controls(B, W + K, rr.del_t, 1235), device=dev).to(dt).contiguous()
        st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); Gs = torch.zeros((B, 6), dtype=dt, device=dev)
        status = torch.zeros((B, K), dtype=torch.int32, device=dev)
        its = torch.zeros((B, K - 2), dtype=torch.int32, device=dev)
        import krod_native as kn
        h.simulate(ctl[:, :W + 2].contiguous()[:, :42], st, Gs, ring=True, use_nn=True)   # 42 = 0 mod 3
        kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(its)))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.simulate(ctl[:, 42:].contiguous(), st, Gs, ring=True, use_nn=True, status=status[:, :K - 2].contiguous(), prev_init=st[2])
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / (K - 2)
        kn.check(h.lib.kr_debug_buffer(h._h, None))
        ii = its.cpu().numpy()
        print(f"{str(dt):14s} predictor={pred}: {el*1e3:.3f} ms/step -> {B/el/1e3:.1f} k rod-steps/s; sweeps/step mean {ii.mean():.2f} max-rod {ii.mean(axis=1).max():.2f} hist {np.bincount(ii.ravel(), minlength=7)[:8]}")
