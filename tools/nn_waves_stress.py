#!/usr/bin/env python3
"""Dev tool (GPU box): randomised comparison of kr_simulate_batch with the MLP on at several wavefronts per rod (2, 4)
with the one-wavefront kernel, and of the long-rod persistent form with the MLP off against one launch per step - grid
sizes, batch sizes, networks, activations, presets, input kinds, step counts, ring / trajectory, fp64 / fp32, iteration caps.
Prints one line per case and a summary; exits non-zero on a mismatch.   python tools/nn_waves_stress.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from cosserat_ode import CosseratRod
from knode import setup_robot
import cosserat_oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2027)
dev = "cuda:0"
MODS = [None, "noair", "short", "damping", "dampstiff", "youngs"]
ACTN = {"elu": "ELU(alpha=1.0)", "tanh": "Tanh()", "relu": "ReLU()", "softplus": "Softplus(beta=1.0, threshold=20.0)"}
bad = 0
ONLY = [int(x) for x in os.environ["KR_STRESS_ONLY"].split(",")] if "KR_STRESS_ONLY" in os.environ else None
for c in range(cases):
    nn = c % 4 != 3                      # every fourth case: MLP off, long rod (persistent form with the states read from HBM)
    mod = MODS[rng.integers(len(MODS))]
    f64 = bool(rng.integers(2))
    dt = torch.float64 if f64 else torch.float32
    ring = bool(rng.integers(2))
    maxit = int(rng.choice([0, 0, 0, 3]))
    kind = rng.choice(["sine", "step", "random"])
    if nn:
        N = int(rng.integers(27, 130))
        B = int(rng.choice([1, 2, 5, 33, 200, 256]))
        T = int(rng.choice([1, 2, 3, 7, 20]))
        layers = [[64, 64], [64], [512], [32, 128], [64, 192], [48, 64]][rng.integers(6)]
        act = ["elu", "tanh", "relu", "softplus"][rng.integers(4)]
    else:
        N = int(rng.integers(260, 420))
        B = int(rng.choice([1, 7, 100, 256]))
        T = int(rng.choice([1, 2, 5, 12]))
    r = CosseratRod(use_fsolve=True); setup_robot(r, mod); r.N = N; r.compute_intermediate_terms()
    if nn:
        mlp = orc.make_mlp([28] + layers + [25], act, seed=int(rng.integers(1 << 16)))
        scale = float(rng.choice([1.0, 0.3]))
        model, params = [], []
        for Wt, b, a in zip(mlp.weights, mlp.biases, mlp.acts):
            model.append("Linear"); params += [Wt * scale, b]
            if a != orc.ACT_NONE: model.append(ACTN[act])
        r.nn_model, r.param_ls, r.nn_path = model, params, "x"
    if kind == "sine":
        ctl = orc.batch_sine_controls(B, T, r.del_t, int(rng.integers(1 << 20)))
    elif kind == "random":
        ctl = 5.0 + 3.0 * rng.uniform(size=(B, T, 4))
    else:
        ctl = np.full((B, T, 4), 5.0); ctl[:, T // 2:, 0] += rng.uniform(0.3, 2.0, size=(B, 1)); ctl[:, T // 2:, 3] += 1.0
    if ONLY is not None and c not in ONLY:
        continue   # (the random stream above is consumed all the same: case c is the same case)
    ctl_t = torch.as_tensor(ctl, device=dev).to(dt).contiguous()
    h = r._native()
    outs = {}
    variants = [("W1", 1, 1), ("W2", 2, 1), ("W4", 4, 1)] if nn else [("step", 0, 0), ("pers", 0, 1)]
    for name, W, pers in variants:
        h.set_option("waves_per_rod", W)
        h.set_option("persistent", pers)
        st = h.new_state(B, dt, n_slots=3 if ring else T + 1)
        h.init_straight(st[0])
        G = torch.zeros((B, 6), dtype=dt, device=dev)
        tip = torch.zeros((B, T, 3), dtype=dt, device=dev)
        status = torch.full((B, T), -1, dtype=torch.int32, device=dev)
        h.simulate(ctl_t, st, G, ring=ring, tip=tip, status=status, maxit=maxit, use_nn=nn)
        torch.cuda.synchronize()
        last = st[T % 3] if ring else st[T]
        outs[name] = (tip.double().cpu().numpy(), status.cpu().numpy(), last[..., :25].double().cpu().numpy(),
                      h.get_option("last_sim_path"), h.get_option("last_waves_per_rod"))
    ref = outs[variants[0][0]]
    tol = (2e-6 if (nn or maxit) else 1e-7) if f64 else 1e-3
    line = f"case {c:3d} {'nn ' + str(layers) + ' ' + act if nn else 'mlp off'} mod={mod} N={N} B={B} T={T} {kind} {'f64' if f64 else 'f32'} ring={int(ring)} maxit={maxit}:"
    ok = True
    for name, W, pers in variants[1:]:
        t, s, l, path, w = outs[name]
        # compare a rod up to the first step either kernel did not converge (what follows starts from different states);
        # statuses must agree when the one-wavefront / per-step kernel converged everywhere and there is no iteration cap -
        # a problem that kernel cannot solve either (exploding network + step input) is reported, not judged
        okmask = np.cumprod((s == 0) & (ref[1] == 0), axis=1).astype(bool)
        dtip = float(np.max(np.abs(t - ref[0])[okmask])) if okmask.any() else 0.0
        hard = bool((ref[1] != 0).any())
        fin = bool(np.isfinite(l).all()) or hard
        same_status = bool(np.array_equal(s, ref[1])) or maxit != 0 or hard
        good = (fin and same_status and dtip < tol and bool(np.all((s >= 0) & (s <= 2)))) or hard
        ok = ok and good
        line += f" {name}(path {path}, W {w}) dtip {dtip:.1e} status_equal {int(np.array_equal(s, ref[1]))} unconv {int((s != 0).sum())}{' (hard: the reference kernel leaves ' + str(int((ref[1] != 0).sum())) + ' steps unconverged itself - not judged)' if hard else ''}{'' if good else ' <-- BAD'};"
    print(line, flush=True)
    if ONLY is not None:
        for name, W, pers in variants:
            t, s, l, path, w = outs[name]
            print(f"    {name}: status counts {dict(zip(*np.unique(s, return_counts=True)))}; first rods' status rows {s[:4].tolist()}; "
                  f"tip[0,-1] {t[0, -1]}", flush=True)
    bad += 0 if ok else 1
print(f"{cases} cases, {bad} bad")
sys.exit(1 if bad else 0)
