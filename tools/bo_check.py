#!/usr/bin/env python3
"""GPU box: MLP-on fp64 simulate with base-only storing sweeps on / off (option nn_base_only_store): the same trajectories.
    python tools/bo_check.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import bench, krod_native as kn, cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
dev = "cuda:0"; dt = torch.float64
worst = 0.0
for name, B, N, T, mk in (("bench workload", 1024, 100, 120, lambda B, T, d: bench.rank_controls(B, 1, 0, T, d)),
                          ("batch_sine 1237", 1024, 100, 120, lambda B, T, d: orc.batch_sine_controls(B, T, d, 1237)),
                          ("N=40 bench", 1024, 40, 120, lambda B, T, d: bench.rank_controls(B, 1, 0, T, d))):
    for seed, scale, act in ((7, 1.0, "elu"), (13, 1.5, "tanh"), (3, 0.3, "softplus"), (21, 2.0, "relu")):
        r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
        mlp = orc.make_mlp([28, 64, 64, 25], act, seed=seed)
        model, params = [], []
        for W_, b_, a_ in zip(mlp.weights, mlp.biases, mlp.acts):
            model.append("Linear"); params += [W_ * scale, b_]
            if a_ != orc.ACT_NONE: model.append({"elu": "ELU(alpha=1.0)", "tanh": "Tanh()", "softplus": "Softplus(beta=1.0, threshold=20.0)", "relu": "ReLU()"}[act])
        r.nn_model, r.param_ls, r.nn_path = model, params, "x"
        h = r._native()
        ctl = torch.as_tensor(mk(B, T, r.del_t), device=dev).to(dt).contiguous()
        res = []
        for on in (1, 0):
            h.set_option("nn_base_only_store", on)
            st = h.new_state(B, dt, n_slots=T + 1); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
            status = torch.zeros((B, T), dtype=torch.int32, device=dev)
            h.simulate(ctl, st, G, ring=False, status=status, use_nn=True)   # (first call of an instantiation)
            st = h.new_state(B, dt, n_slots=T + 1); h.init_straight(st[0]); G.zero_()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h.simulate(ctl, st, G, ring=False, status=status, use_nn=True)
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / T
            res.append((st.cpu().numpy().copy(), int((status != 0).sum()), el, h.get_option("last_sim_path")))
        # reference: the same with a tolerance of 1e-11 (base-only sweeps off)
        h.set_option("nn_base_only_store", 0)
        st = h.new_state(B, dt, n_slots=T + 1); h.init_straight(st[0]); G.zero_()
        h.simulate(ctl, st, G, ring=False, status=status, use_nn=True, tol=1e-11, maxit=30)
        torch.cuda.synchronize()
        ref = st.cpu().numpy().copy()
        def err(x):
            return float((np.sqrt(((x[1:] - ref[1:])[..., :25] ** 2).sum(axis=(2, 3))) / np.sqrt((ref[1:][..., :25] ** 2).sum(axis=(2, 3)))).max())
        def tip_err(x):
            """the contract's figure: rel L2 of the tip trajectory (slots 12..14 of the last grid point) per rod, worst rod"""
            d = x[1:, :, -1, 12:15] - ref[1:, :, -1, 12:15]
            return float((np.sqrt((d ** 2).sum(axis=(0, 2))) / np.sqrt((ref[1:, :, -1, 12:15] ** 2).sum(axis=(0, 2)))).max())
        (a, ba, ta, pa), (b, bb, tb, pb) = res
        print(f"   against a run at tol 1e-11 ({int((status != 0).sum())} unconverged): on {err(a):.2e}, off {err(b):.2e}")
        # fp32 sweeps, both ways, against the same fp64 reference
        ctl32 = ctl.float().contiguous()
        e32 = []
        for on in (1, 0):
            h.set_option("nn_base_only_store", on)
            for rep in range(2):
                st32 = h.new_state(B, torch.float32, n_slots=T + 1); h.init_straight(st32[0]); G32 = torch.zeros((B, 6), dtype=torch.float32, device=dev)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                h.simulate(ctl32, st32, G32, ring=False, status=status, use_nn=True)
                torch.cuda.synchronize(); el32 = (time.perf_counter() - t0) / T
            x32 = st32.double().cpu().numpy()
            e32.append((err(x32), el32, int((status != 0).sum()), tip_err(x32)))
        print(f"   fp32 sweeps: on {e32[0][1]*1e3:.3f} ms/step err {e32[0][0]:.2e} TIP {e32[0][3]:.2e} ({e32[0][2]} unconverged), off {e32[1][1]*1e3:.3f} ms/step err {e32[1][0]:.2e} TIP {e32[1][3]:.2e} ({e32[1][2]} unconverged)"
              f"   [fp64 TIP: on {tip_err(a):.2e} off {tip_err(b):.2e}]")
        # per rod and step, over the whole state (slots 0..24 of every grid point)
        num = np.sqrt(((a[1:] - b[1:])[..., :25] ** 2).sum(axis=(2, 3)))
        den = np.sqrt((b[1:][..., :25] ** 2).sum(axis=(2, 3)))
        rel = float((num / den).max())
        worst = max(worst, rel)
        print(f"{name:16s} seed {seed:2d} x{scale} {act:8s}: on {ta*1e3:.3f} ms/step, off {tb*1e3:.3f} ms/step (path {pa}); unconverged {ba} / {bb}; "
              f"worst state rel L2 difference over rods and steps {rel:.2e}", flush=True)
print("worst:", worst)
