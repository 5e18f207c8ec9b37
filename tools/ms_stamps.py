#!/usr/bin/env python3
"""Dev tool (GPU box): per-phase cycle counters of the persistent multiple-shooting kernel.
Needs the diagnostic library: KR_LIB_PATH=knode-cosserat_amd/lib/dbg/libknode_rod.so"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc, krod_native as kn
from cosserat_ode import CosseratRod
from knode import setup_robot
B, N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 100, int(os.environ.get("KR_STAMP_T", "120"))
NN = len(sys.argv) > 2 and sys.argv[2] == "nn"   # MLP 28 -> 64 -> 64 -> 25 inside the sweeps
if NN: T = 40
dev = "cuda:0"; dt = torch.float64
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
if NN:
    mlp = orc.make_mlp([28, 64, 64, 25], "elu", seed=7)
    if os.environ.get("KR_STAMP_BF16W"):  # probe: weights exactly representable in bf16 (the JVP operands then carry no rounding)
        def bf16r(a):
            u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
            u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
            return u.view(np.float32)
        mlp.weights = [bf16r(w) for w in mlp.weights]
    if os.environ.get("KR_STAMP_PBLIND"):  # probe: a network that does not read p (the condensed Jacobian has no columns for p)
        mlp.weights[0] = mlp.weights[0].copy(); mlp.weights[0][:, 0:3] = 0.0
    model, params = [], []
    for W_, b_, a_ in zip(mlp.weights, mlp.biases, mlp.acts):
        model.append("Linear"); params += [W_, b_]
        if a_ != orc.ACT_NONE: model.append("ELU(alpha=1.0)")
    r.nn_model, r.param_ls, r.nn_path = model, params, "x"
h = r._native(); h.set_option("ms_mode", 1); h.set_option("persistent", 1)
WARM = int(os.environ.get("KR_STAMP_WARM", "21"))
if os.environ.get("KR_STAMP_KEEP"): h.set_option("keep_predictor", 1)   # steady state: the timed call resumes the predictor
dbg = torch.zeros((B, 24), dtype=torch.int64, device=dev)
kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(dbg)))
ctl_all = torch.as_tensor(orc.batch_sine_controls(B, WARM + T, r.del_t, 1235), device=dev).contiguous()
ctl = ctl_all[:, WARM:].contiguous() if os.environ.get("KR_STAMP_KEEP") else ctl_all[:, :T].contiguous()
for pred in (7, 8):
    h.set_option("predictor", pred)
    st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
    h.simulate(ctl_all[:, :WARM].contiguous(), st, G, ring=True, use_nn=NN)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.simulate(ctl, st, G, ring=True, prev_init=st[2], use_nn=NN)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    d = dbg.cpu().numpy().astype(np.float64)
    tot, sw, al, pr, its = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4]
    print(f"pred={pred}: wall {el/T*1e6:.1f} us/step; s_memtime ticks per step: total mean {tot.mean()/T:.0f} max {tot.max()/T:.0f} | "
          f"sweep {sw.mean()/T:.0f} alg {al.mean()/T:.0f} prep {pr.mean()/T:.0f} | its/step mean {its.mean()/T:.2f} max {its.max()/T:.2f} | "
          f"per-iteration sweep {sw.sum()/its.sum():.0f} alg {al.sum()/its.sum():.0f} ticks; tick rate {tot.max()/el/1e6:.1f} MHz")
    print(f"   sweeps accepted by the residual test, per step: {dbg[:, 15].cpu().numpy().view(np.float64).mean()/T:.3f}")
    print(f"   algebra per iteration: hand-over {d[:,8].sum()/its.sum():.0f} chain {d[:,9].sum()/its.sum():.0f} solve {d[:,10].sum()/its.sum():.0f} update {d[:,11].sum()/its.sum():.0f}")
    dn = dbg[:, 5:8].cpu().numpy().view(np.float64)
    its_rod = its / T
    order = np.argsort(-its_rod)
    print("   worst rods (its/step, last step dn1 dn2 dn3):")
    for b in order[:6]: print(f"     rod {b}: {its_rod[b]:.2f}  {dn[b,0]:.1e} {dn[b,1]:.1e} {dn[b,2]:.1e}  mean order {d[b,12]/T:.2f} last {int(d[b,13])} retries {int(d[b,14])}")
    em = dbg[:, 16:24].cpu().numpy().view(np.float64)
    for b in order[:4]: print("     rod", b, "prediction errors by order (last step):", " ".join(f"{x:.1e}" for x in em[b]))
    print("   order histogram (last step):", np.bincount(d[:,13].astype(int), minlength=8), " retries total", int(d[:,14].sum()))
    print("   best rods:")
    for b in order[-3:]: print(f"     rod {b}: {its_rod[b]:.2f}  {dn[b,0]:.1e} {dn[b,1]:.1e} {dn[b,2]:.1e}")
    print("   dn1 quantiles", np.quantile(dn[:,0],[0.5,0.9,0.99,1.0]), " dn2", np.quantile(dn[:,1],[0.5,0.9,0.99,1.0]), " dn3", np.quantile(dn[:,2],[0.5,0.9,0.99,1.0]))
