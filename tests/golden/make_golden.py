#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); the GPU box never
sees the reference, only the ``.npz`` files this script writes.  Fixtures are
data only: inputs and the reference's outputs on them.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [names...]

Fixture list follows SURVEY.md section 8c (F1..F8).
"""
import os
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = "/root/reference/knode_cosserat"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))

import numpy as np
import torch
import torch.nn as nn

import cosserat_ode as ref_ode  # noqa: E402  (reference)
import cosserat_ode_torch as ref_torch  # noqa: E402
import knode as ref_knode  # noqa: E402
import physics_controls as ref_ctl  # noqa: E402
from Utils.transformations import quaternion_to_euler as ref_q2e  # noqa: E402

import cosserat_oracle as orc  # only for make_mlp / batch_sine_controls (input generators)

torch.set_num_threads(1)
MODS = [None, "noair", "nsw", "short", "damping", "dampstiff", "lengthstiff", "youngs"]
ACT_MODULE = {"tanh": nn.Tanh, "softplus": nn.Softplus, "relu": nn.ReLU, "elu": nn.ELU}


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def np_robot(mod="default", N=10, use_fsolve=True):
    r = ref_ode.CosseratRod(use_fsolve=use_fsolve)
    if mod != "default":
        ref_knode.setup_robot(r, mod)
    r.N = N
    r.compute_intermediate_terms()
    return r


def torch_module_list(mlp):
    """nn.ModuleList with the layer order the reference walks by str()."""
    mods = []
    names = {v: k for k, v in orc._ACT_BY_NAME.items() if k != "identity"}
    for W, b, act in zip(mlp.weights, mlp.biases, mlp.acts):
        lin = nn.Linear(W.shape[1], W.shape[0])
        with torch.no_grad():
            lin.weight.copy_(torch.tensor(W))
            lin.bias.copy_(torch.tensor(b))
        mods.append(lin)
        if act != orc.ACT_NONE:
            mods.append(ACT_MODULE[names[act]]())
    return nn.ModuleList(mods)


def inject_nn(robot, mlp):
    """What physics_train.py:104-110 does to switch the NumPy rod to NN mode."""
    ml = torch_module_list(mlp)
    robot.nn_model = ml
    robot.param_ls = [t.detach().cpu().numpy() for _, t in ml.state_dict().items()]
    robot.nn_path = "whatever"
    robot.nn_input_history = mlp.history


def mlp_arrays(prefix, mlp):
    d = {}
    for k, (W, b) in enumerate(zip(mlp.weights, mlp.biases)):
        d[f"{prefix}_W{k}"] = W
        d[f"{prefix}_b{k}"] = b
    d[f"{prefix}_acts"] = np.array(mlp.acts, dtype=np.int32)
    d[f"{prefix}_history"] = np.array(int(mlp.history))
    return d


def sample_rows(Q, seed):
    """Rows with the scale of a real trajectory: states of a short reference
    run (N=10, sine) jittered; quaternion deliberately not normalised."""
    rng = np.random.default_rng(seed)
    r = np_robot(None, 10)
    ctl = ref_ctl.calc_controls("sine", 1.0, r.del_t, 24)
    traj = ref_knode.simulate(r, ctl)
    ys, yhs, zhs = [], [], []
    for _ in range(Q):
        t = rng.integers(2, traj.shape[0])
        j = rng.integers(0, 10)
        y = traj[t, :19, j].copy()
        y *= 1 + 0.2 * rng.standard_normal(19)
        y += 1e-3 * rng.standard_normal(19)
        y[3:7] *= rng.uniform(0.7, 1.4)
        ys.append(y)
        yhs.append(traj[t, 25:44, j] * (1 + 0.1 * rng.standard_normal(19)))
        zhs.append(traj[t, 44:50, j] * (1 + 0.1 * rng.standard_normal(6)))
    tens = 5 + 2 * rng.random((Q, 4))
    return np.array(ys), np.array(yhs), np.array(zhs), tens


NN_VARIANTS = [
    ("elu64", [28, 64, 25], "elu", False),
    ("hist64", [53, 64, 25], "elu", True),
    ("tanh6464", [28, 64, 64, 25], "tanh", False),
    ("softplus6464", [28, 64, 64, 25], "softplus", False),
    ("relu6464", [28, 64, 64, 25], "relu", False),
    ("elu6464", [28, 64, 64, 25], "elu", False),
]


def gen_ode_kat():
    """F1: CosseratRod.ODE on random rows x presets x NN variants."""
    Q = 48
    y, yh, zh, tens = sample_rows(Q, 0)
    out = {"y": y, "yh": yh, "zh": zh, "tensions": tens}
    for mod in ["default"] + MODS:
        r = np_robot(mod, 10)
        res = np.zeros((Q, 25))
        for i in range(Q):
            tf = tens[i] @ r.tendon_dirs
            ys, z = r.ODE(y[i].copy(), yh[i], zh[i], tf)
            res[i] = np.concatenate([ys, z])
        out[f"phys_{mod}"] = res
    for k, (name, sizes, act, hist) in enumerate(NN_VARIANTS):
        mlp = orc.make_mlp(sizes, act, seed=100 + k, history=hist)
        # scale the weights up so the correction is not negligible next to the physics
        mlp.weights = [w * 3 for w in mlp.weights]
        r = np_robot(None, 10)
        inject_nn(r, mlp)
        res = np.zeros((Q, 25))
        for i in range(Q):
            tf = tens[i] @ r.tendon_dirs
            ys, z = r.ODE(y[i].copy(), yh[i], zh[i], tf)
            res[i] = np.concatenate([ys, z])
        out[f"nn_{name}"] = res
        out.update(mlp_arrays(f"mlp_{name}", mlp))
    save("ode_kat", **out)


def gen_ode_torch_kat():
    """F2: the torch twin, serial ODE and ODE_parallel, fp32."""
    Q = 48
    y, yh, zh, tens = sample_rows(Q, 0)
    out = {}
    for name, sizes, act, hist in [NN_VARIANTS[0], NN_VARIANTS[1], NN_VARIANTS[5]]:
        k = [v[0] for v in NN_VARIANTS].index(name)
        mlp = orc.make_mlp(sizes, act, seed=100 + k, history=hist)
        mlp.weights = [w * 3 for w in mlp.weights]
        rob = ref_torch.CosseratRodTorch("cpu", 64, nn_input_history=hist)
        ref_knode.setup_robot(rob, None)
        rob.nn_models = torch_module_list(mlp)
        ty, tyh, tzh = (torch.tensor(a).float() for a in (y, yh, zh))
        tf = torch.tensor(tens).float() @ rob.tendon_dirs
        for use_nn in (False, True):
            rob.use_nn = use_nn
            with torch.no_grad():
                dys, z = rob.ODE_parallel(ty, tyh, tzh, tf)
                par = torch.cat([dys, z], 1).numpy()
                ser = np.zeros((Q, 25), np.float32)
                for i in range(Q):
                    a, b = rob.ODE(ty[i].clone(), tyh[i], tzh[i], tf[i])
                    ser[i] = torch.cat([a, b]).numpy()
            out[f"par_{name}_{int(use_nn)}"] = par
            out[f"ser_{name}_{int(use_nn)}"] = ser
    save("ode_torch_kat", **out)


def converged_state(r, T, ctl_fn):
    """Run the reference T steps and return (y, z, y_prev, z_prev, G, tensions)
    from which one more residual can be evaluated."""
    ctl = ctl_fn(T + 1)
    traj = ref_knode.simulate(r, ctl)
    y, z = traj[T - 1, :19].copy(), traj[T - 1, 19:25].copy()
    yp, zp = traj[T - 2, :19].copy(), traj[T - 2, 19:25].copy()
    return y, z, yp, zp, traj[T - 1, 7:13, 0].copy(), np.array(ctl[T - 1], float)


def gen_residual_kat():
    """F3: getResidualEuler / getResidualRK4 6-vectors and the mutated y, z."""
    rng = np.random.default_rng(1)
    out = {}
    for N, mod in [(10, None), (20, None), (100, None), (10, "default"), (40, "default")]:
        r = np_robot(mod, N)
        y, z, yp, zp, G0, tens = converged_state(
            r, 6, lambda T: ref_ctl.calc_controls("sine", 1.0, r.del_t, T))
        yh = r.c1 * y + r.c2 * yp
        zh = r.c1 * z + r.c2 * zp
        yh_int = 0.5 * (yh[:, :-1] + yh[:, 1:])
        zh_int = 0.5 * (zh[:, :-1] + zh[:, 1:])
        r.tendon_tensions = tens
        tag = f"N{N}_{mod}"
        out[f"{tag}_y"], out[f"{tag}_z"], out[f"{tag}_yp"], out[f"{tag}_zp"] = y, z, yp, zp
        out[f"{tag}_tens"] = tens
        Gs = G0[None, :] * (1 + 0.05 * rng.standard_normal((3, 6))) + 1e-3 * rng.standard_normal((3, 6))
        out[f"{tag}_G"] = Gs
        for scheme, fn in (("euler", r.getResidualEuler), ("rk4", r.getResidualRK4)):
            rs, ys, zs = [], [], []
            for G in Gs:
                yy, zz = y.copy(), z.copy()
                with np.errstate(all="ignore"):
                    rs.append(fn(G, yy, zz, yh, yh_int, zh, zh_int))
                ys.append(yy)
                zs.append(zz)
            out[f"{tag}_{scheme}_r"] = np.array(rs)
            out[f"{tag}_{scheme}_y"] = np.array(ys)
            out[f"{tag}_{scheme}_z"] = np.array(zs)
    save("residual_kat", **out)


class FsolveSpy:
    """Wraps scipy's fsolve so that ier / nfev of every step are recorded;
    the reference discards them (knode.py:89)."""

    def __init__(self):
        from scipy.optimize import fsolve
        self._f = fsolve
        self.ier, self.nfev = [], []

    def __call__(self, fun, x0, args=()):
        x, info, ier, _ = self._f(fun, x0, args=args, full_output=True)
        self.ier.append(ier)
        self.nfev.append(info["nfev"])
        return x


def run_sim(r, ctl, scheme="euler"):
    spy = FsolveSpy()
    old = ref_knode.fsolve
    ref_knode.fsolve = spy
    if scheme == "rk4":
        r_euler = r.getResidualEuler
        r.getResidualEuler = r.getResidualRK4  # knode.simulate hard-codes the Euler residual (knode.py:89)
    try:
        with np.errstate(all="ignore"):
            traj = ref_knode.simulate(r, ctl)
    finally:
        ref_knode.fsolve = old
        if scheme == "rk4":
            r.getResidualEuler = r_euler
    return traj, np.array(spy.ier), np.array(spy.nfev)


def gen_sim_cfg1():
    """F4: BASELINE config 1 - single rod, N=20, 200 steps, tensions [6,5,5,6]."""
    r = np_robot(None, 20)
    ctl = [[6.0, 5.0, 5.0, 6.0]] * 200
    t0 = time.time()
    traj, ier, nfev = run_sim(r, ctl)
    print(f"  cfg1: {200/(time.time()-t0):.1f} rod-steps/s, mean nfev {nfev.mean():.1f}")
    save("sim_cfg1", ctl=np.array(ctl), tip=traj[:, :3, -1], every10=traj[::10, :25], last=traj[-1],
         ier=ier, nfev=nfev)


def gen_sim_n100():
    """F5a: N=100, T=50, sine(1.0); plus a small cfg2-style random-phase batch."""
    r = np_robot(None, 100)
    ctl = ref_ctl.calc_controls("sine", 1.0, r.del_t, 50)
    traj, ier, nfev = run_sim(r, ctl)
    out = dict(ctl=np.array(ctl), tip=traj[:, :3, -1], every10=traj[::10, :25], last=traj[-1], ier=ier, nfev=nfev)
    B, T = 6, 24
    ctl_b = orc.batch_sine_controls(B, T, r.del_t, 1234)
    tips, iers = [], []
    for b in range(B):
        tr, ie, _ = run_sim(r, ctl_b[b])
        tips.append(tr[:, :3, -1])
        iers.append(ie)
    out.update(batch_ctl=ctl_b, batch_tip=np.array(tips), batch_ier=np.array(iers))
    save("sim_n100", **out)


def gen_sim_n400():
    """F5b: N=400, T=12, sine(1.0)."""
    r = np_robot(None, 400)
    ctl = ref_ctl.calc_controls("sine", 1.0, r.del_t, 12)
    traj, ier, nfev = run_sim(r, ctl)
    save("sim_n400", ctl=np.array(ctl), tip=traj[:, :3, -1], last=traj[-1, :25], ier=ier, nfev=nfev)


def gen_sim_misc():
    """Presets, default parameters, step / random inputs, RK4 - short runs at small N."""
    out = {}
    for mod in MODS[1:] + ["default"]:
        r = np_robot(mod, 10)
        ctl = ref_ctl.calc_controls("sine", 1.0, r.del_t, 16)
        traj, ier, _ = run_sim(r, ctl)
        out[f"mod_{mod}_ctl"] = np.array(ctl)
        out[f"mod_{mod}_traj"] = traj[:, :25]
        out[f"mod_{mod}_ier"] = ier
    r = np_robot(None, 10)
    for kind, arg, T in (("step", 1.0, 40), ("random", 3.0, 20)):
        ctl = ref_ctl.calc_controls(kind, arg, r.del_t, T)
        traj, ier, _ = run_sim(r, ctl)
        out[f"{kind}_ctl"] = np.array(ctl)
        out[f"{kind}_traj"] = traj[:, :25]
        out[f"{kind}_ier"] = ier
    # full 50-row output once, to pin the [y; z; yh; zh] stacking and the [:-1] drop
    ctl = ref_ctl.calc_controls("sine", 2.0, r.del_t, 8)
    traj, ier, _ = run_sim(r, ctl)
    out["full50_ctl"], out["full50_traj"], out["full50_ier"] = np.array(ctl), traj, ier
    # RK4 (stable only for fine grids with the experimental preset)
    r = np_robot(None, 40)
    ctl = ref_ctl.calc_controls("sine", 1.0, r.del_t, 10)
    traj, ier, _ = run_sim(r, ctl, scheme="rk4")
    out["rk4_ctl"], out["rk4_traj"], out["rk4_ier"] = np.array(ctl), traj[:, :25], ier
    save("sim_misc", **out)


def gen_sim_nn():
    """F6: forward simulation with the residual MLP switched on."""
    out = {}
    for name, sizes, act, hist, N, T in (("elu64", [28, 64, 25], "elu", False, 10, 30),
                                         ("elu6464", [28, 64, 64, 25], "elu", False, 20, 20),
                                         ("hist64", [53, 64, 25], "elu", True, 10, 20)):
        mlp = orc.make_mlp(sizes, act, seed=0, history=hist)
        if hist:
            # history inputs are ~c1 * state, 40x larger: keep the correction small enough that the
            # reference's own solve stays convergent
            mlp.weights[0] = (mlp.weights[0] * 0.02).astype(np.float32)
        r = np_robot(None, N)
        inject_nn(r, mlp)
        ctl = ref_ctl.calc_controls("sine", 1.0, r.del_t, T)
        traj, ier, nfev = run_sim(r, ctl)
        out[f"{name}_ctl"], out[f"{name}_traj"], out[f"{name}_ier"] = np.array(ctl), traj[:, :25], ier
        out[f"{name}_N"] = np.array(N)
        out.update(mlp_arrays(f"mlp_{name}", mlp))
    save("sim_nn", **out)


def four_term_loss(pred, target, kp_pred, kp_tgt):
    """The loss of physics_train.py:252-259 / :345-352 built from the
    reference's own pieces (nn.MSELoss + Utils.transformations)."""
    mse = nn.MSELoss()
    return (mse(pred[:3][:, kp_pred], target[:3, kp_tgt])
            + mse(pred[7:19][:, kp_pred], target[7:19, kp_tgt])
            + mse(ref_q2e(pred[3:7][:, kp_pred]), ref_q2e(target[3:7, kp_tgt]))
            + mse(pred[19:][:, kp_pred], target[19:, kp_tgt - 1]))


def gen_train_step():
    """F7: one slow-path epoch and one --fast epoch of physics_train.py on a
    fixed trajectory and fixed weights: predictions, loss, gradients and the
    weights after Adam(lr=1e-2) + clamp(min=0)."""
    out = {}
    r = np_robot(None, 10)
    ctl = np.array(ref_ctl.calc_controls("sine", 2.0, r.del_t, 30))
    traj_np = ref_knode.simulate(r, ctl)[:, :25]
    traj = torch.tensor(traj_np).float()
    controls = torch.tensor(ctl).float()
    out["traj"], out["controls"] = traj.numpy(), controls.numpy()
    for H in (64,):
        mlp = orc.make_mlp([28, H, 25], "elu", seed=7)
        out.update(mlp_arrays("mlp", mlp))
        for path in ("slow", "fast"):
            rob = ref_torch.CosseratRodTorch("cpu", H)
            ref_knode.setup_robot(rob, "damping")  # a preset that differs from the data generator, as in training
            rob.nn_models = torch_module_list(mlp)
            rob.use_nn = True
            opt = torch.optim.Adam(rob.nn_models.parameters(), lr=1e-2, weight_decay=0)
            loss = 0
            preds = []
            if path == "slow":
                kp = torch.tensor([2, 6, 9])
                for t in range(29):
                    y, z = traj[t, :19], traj[t, 19:]
                    yp, zp = (y, z) if t == 0 else (traj[t - 1, :19], traj[t - 1, 19:])
                    rob.y, rob.z = y, z
                    rob.tendon_tensions = controls[t]
                    rob.residualArgs["yh"] = rob.c1 * y + rob.c2 * yp
                    rob.residualArgs["zh"] = rob.c1 * z + rob.c2 * zp
                    grow = rob.getNextSegmentEuler(traj[t + 1].clone())
                    preds.append(grow.detach().numpy())
                    loss = loss + four_term_loss(grow, traj[t + 1], kp, kp)
            else:
                kp = np.array([3, 5, 7, 9])
                ys, zs = traj[:29, :19], traj[:29, 19:]
                yps, zps = torch.cat((ys[:1], ys[:-1])), torch.cat((zs[:1], zs[:-1]))
                grows = rob.parallelGetNextSegmentEuler(traj[1:30], kp, {
                    "yh": rob.c1 * ys + rob.c2 * yps, "zh": rob.c1 * zs + rob.c2 * zps,
                    "tendon_tensions": controls[:29]})
                for t in range(29):
                    preds.append(grows[t].detach().numpy())
                    loss = loss + four_term_loss(grows[t], traj[t + 1], torch.arange(4), torch.tensor(kp))
            loss = loss / 29
            opt.zero_grad()
            loss.backward()
            grads = [p.grad.detach().numpy().copy() for p in rob.nn_models.parameters()]
            opt.step()
            with torch.no_grad():
                for nm, p in rob.nn_models.named_parameters():
                    if "weight" in nm and "layer1" not in nm:
                        p.clamp_(min=0)
            post = [p.detach().numpy().copy() for p in rob.nn_models.parameters()]
            out[f"{path}_pred"] = np.array(preds)
            out[f"{path}_loss"] = np.array(loss.item())
            out[f"{path}_kp"] = np.asarray(kp)
            for i, (gk, pk) in enumerate(zip(grads, post)):
                out[f"{path}_grad{i}"] = gk
                out[f"{path}_post{i}"] = pk
        # no-NN self-consistency: the predictor reproduces the next state
        rob.use_nn = False
        with torch.no_grad():
            t = 11
            rob.tendon_tensions = controls[t]
            rob.residualArgs["yh"] = rob.c1 * traj[t, :19] + rob.c2 * traj[t - 1, :19]
            rob.residualArgs["zh"] = rob.c1 * traj[t, 19:] + rob.c2 * traj[t - 1, 19:]
            out["nonn_pred_t11"] = rob.getNextSegmentEuler(traj[t + 1].clone()).numpy()
    save("train_step", **out)


def gen_checkpoint():
    """Checkpoint interchange: a file written exactly as physics_train.py:284-288 writes it
    (torch.save({'robot': robot, 'dtw': ..., 'loss': ..., 'optim': ...})) with the REFERENCE classes,
    plus what the reference computes with that robot.  The .pth holds pickled attribute values only."""
    torch.manual_seed(3)
    rob = ref_torch.CosseratRodTorch("cpu", 32)
    ref_knode.setup_robot(rob, "damping")
    opt = torch.optim.Adam(rob.nn_models.parameters(), lr=1e-2)
    path = os.path.join(HERE, "ref_checkpoint.pth")
    torch.save({"robot": rob, "dtw": [[1.5]], "loss": [0.25], "optim": opt.state_dict()}, path)
    Q = 16
    y, yh, zh, tens = sample_rows(Q, 4)
    ty, tyh, tzh = (torch.tensor(a).float() for a in (y, yh, zh))
    tf = torch.tensor(tens).float() @ rob.tendon_dirs
    with torch.no_grad():
        dys, z = rob.ODE_parallel(ty, tyh, tzh, tf)
    # the NumPy class reading the same file (cosserat_ode.py:81-88 uses map_location 'mps'; cpu here)
    nn_model = torch.load(path, map_location="cpu", weights_only=False)["robot"].nn_models
    params = {k: v.detach().numpy() for k, v in nn_model.state_dict().items()}
    save("checkpoint", y=y, yh=yh, zh=zh, tens=tens, par=torch.cat([dys, z], 1).numpy(),
         L=np.float64(rob.L), del_t=np.float64(rob.del_t), E=np.float64(rob.E), Bbt=rob.Bbt.numpy(),
         layer_strings=np.array([str(l) for l in nn_model]), **{"p_" + k: v for k, v in params.items()})


def gen_estimate_state():
    """knode_cosserat_realworld/estimate_state.py run on poses taken from a simulated
    trajectory (N = 10 grid points, the only size its hard-coded index 9 is meant for) and on an N = 12
    variant that exercises the wrap-around of that quirk."""
    rw = "/root/reference/knode_cosserat_realworld"
    import importlib.util
    argv, sys.argv = sys.argv, ["x"]
    sys.path.insert(0, rw)
    try:
        spec = importlib.util.spec_from_file_location("ref_estimate_state", os.path.join(rw, "estimate_state.py"))
        es = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(es)
    finally:
        sys.path.remove(rw)
        sys.argv = argv
    out = {}
    for N in (10, 12):
        r = np_robot(None, N)
        T = 40
        ctl = np.array(ref_ctl.calc_controls("sine", 1.0, r.del_t, T))
        traj = ref_knode.simulate(r, ctl)[:, :25]
        rng = np.random.default_rng(N)
        data = traj[:, :7, :] + 1e-4 * rng.standard_normal(traj[:, :7, :].shape)  # measurement noise
        r2 = np_robot(None, N)
        est = es.estimate_state(data.copy(), ctl.copy(), r2)
        out[f"N{N}_data"] = data
        out[f"N{N}_ctl"] = ctl
        out[f"N{N}_est"] = est
        out[f"N{N}_vstar_after"] = np.array(r2.vstar, dtype=np.float64)
    save("estimate_state", **out)


def gen_sim_more():
    """Round-2 additions: (a) the reference's DEFAULT network 28 -> 512 -> 25 (cosserat_ode_torch.py:60-62,
    physics_train.py:47) inside simulate; (b) BASELINE cfg5 inputs: N=400 under calc_controls('sine', P) for
    P in {0.5, 2, 3} (P=1: sim_n400.npz); (c) the use_fsolve=False branch of knode.simulate (knode.py:91-94,
    L-BFGS-B on the sum of squares); (d) the torch twin's differentiable full sweep getResidualEuler(G)
    (cosserat_ode_torch.py:325-367), NN off and on."""
    out = {}
    # "elu512": weights exactly as the reference initialises them - with 512 all-positive hidden units the untrained
    # correction moves the tip by 85 % and plain Newton from the warm start diverges from step 10 on (hybrd's trust
    # region copes): pins the damped fallback of the solver.  "elu512n24": the same network at 0.3 x the weights.
    for name, N, T, wscale in (("elu512", 10, 30, 1.0), ("elu512n24", 24, 12, 0.3)):
        mlp = orc.make_mlp([28, 512, 25], "elu", seed=3)
        mlp.weights = [(w * wscale).astype(np.float32) for w in mlp.weights]
        r = np_robot(None, N)
        inject_nn(r, mlp)
        ctl = ref_ctl.calc_controls("sine", 1.0, r.del_t, T)
        traj, ier, nfev = run_sim(r, ctl)
        out[f"{name}_ctl"], out[f"{name}_traj"], out[f"{name}_ier"] = np.array(ctl), traj[:, :25], ier
        out[f"{name}_N"] = np.array(N)
        out.update(mlp_arrays(f"mlp_{name}", mlp))
    for P in (0.5, 2.0, 3.0):
        r = np_robot(None, 400)
        ctl = ref_ctl.calc_controls("sine", P, r.del_t, 8)
        traj, ier, nfev = run_sim(r, ctl)
        tag = f"n400_P{P}".replace(".", "_")
        out[f"{tag}_ctl"], out[f"{tag}_tip"], out[f"{tag}_last"], out[f"{tag}_ier"] = (
            np.array(ctl), traj[:, :3, -1], traj[-1, :25], ier)
    # (c) minimize branch: the reference's own driver loop, unmodified
    r = np_robot(None, 10, use_fsolve=False)
    ctl = ref_ctl.calc_controls("sine", 1.0, r.del_t, 12)
    with np.errstate(all="ignore"):
        traj = ref_knode.simulate(r, ctl)
    out["lbfgs_ctl"], out["lbfgs_traj"] = np.array(ctl), traj
    # (d) torch getResidualEuler(G): state of a short run, G slightly off the root
    rn = np_robot(None, 10)
    y, z, yp, zp, G0, tens = converged_state(rn, 6, lambda T: ref_ctl.calc_controls("sine", 1.0, rn.del_t, T))
    mlp = orc.make_mlp([28, 64, 25], "elu", seed=11)
    mlp.weights = [w * 3 for w in mlp.weights]
    out.update(mlp_arrays("mlp_tres", mlp))
    out["tres_y"], out["tres_z"], out["tres_yp"], out["tres_zp"], out["tres_tens"] = y, z, yp, zp, tens
    Gs = np.stack([G0, G0 * 1.03 + 1e-3])
    out["tres_G"] = Gs
    for use_nn in (0, 1):
        rob = ref_torch.CosseratRodTorch("cpu", 64)
        ref_knode.setup_robot(rob, None)
        rob.nn_models = torch_module_list(mlp)
        rob.use_nn = bool(use_nn)
        ty, tz = torch.tensor(y).float(), torch.tensor(z).float()
        typ, tzp = torch.tensor(yp).float(), torch.tensor(zp).float()
        vals, rods, ys = [], [], []
        for G in Gs:
            rob.y, rob.z = ty.clone(), tz.clone()
            rob.tendon_tensions = torch.tensor(tens).float()
            rob.residualArgs["yh"] = rob.c1 * ty + rob.c2 * typ
            rob.residualArgs["zh"] = rob.c1 * tz + rob.c2 * tzp
            with torch.no_grad():
                tot, full = rob.getResidualEuler(torch.tensor(G).float())
            vals.append(float(tot))
            rods.append(full.numpy())
            ys.append(rob.y.numpy().copy())
        out[f"tres_val_{use_nn}"] = np.array(vals)
        out[f"tres_full_{use_nn}"] = np.array(rods)
        out[f"tres_yafter_{use_nn}"] = np.array(ys)
    save("sim_more", **out)


def gen_tres_grad():
    """a14 with autograd: gradients of  L = total_residual + sum(full_rod * Wgt)  (cosserat_ode_torch.py:325-367) with
    respect to the guessed base wrench G and every MLP parameter, on the inputs of the `tres_*` entries of
    sim_more.npz (same state, same network; use_nn 0 and 1)."""
    g = np.load(os.path.join(HERE, "sim_more.npz"))
    mlp = orc.Mlp([g[f"mlp_tres_W{k}"] for k in range(2)], [g[f"mlp_tres_b{k}"] for k in range(2)],
                  [int(a) for a in g["mlp_tres_acts"]], bool(g["mlp_tres_history"]))
    y, z, yp, zp, tens = g["tres_y"], g["tres_z"], g["tres_yp"], g["tres_zp"], g["tres_tens"]
    G = g["tres_G"][1]
    rng = np.random.default_rng(21)
    Wgt = rng.standard_normal((25, y.shape[1])).astype(np.float32)
    out = {"Wgt": Wgt}
    for use_nn in (0, 1):
        rob = ref_torch.CosseratRodTorch("cpu", 64)
        ref_knode.setup_robot(rob, None)
        rob.nn_models = torch_module_list(mlp)
        rob.use_nn = bool(use_nn)
        ty, tz = torch.tensor(y).float(), torch.tensor(z).float()
        typ, tzp = torch.tensor(yp).float(), torch.tensor(zp).float()
        rob.y, rob.z = ty.clone(), tz.clone()
        rob.tendon_tensions = torch.tensor(tens).float()
        rob.residualArgs["yh"] = rob.c1 * ty + rob.c2 * typ
        rob.residualArgs["zh"] = rob.c1 * tz + rob.c2 * tzp
        Gt = torch.tensor(G).float().requires_grad_(True)
        tot, full = rob.getResidualEuler(Gt)
        L = tot + (full * torch.tensor(Wgt)).sum()
        L.backward()
        out[f"L_{use_nn}"] = np.array(float(L))
        out[f"dG_{use_nn}"] = Gt.grad.numpy().copy()
        if use_nn:
            for k, prm in enumerate(rob.nn_models.parameters()):
                out[f"dparam{k}"] = prm.grad.numpy().copy()
    # ODE_parallel with autograd into its INPUTS (cosserat_ode_torch.py:217-322; the graph there is intact): rows and
    # networks of ode_kat.npz / ode_torch_kat.npz, L = sum(dys * Wd) + sum(z * Wz)
    Q = 48
    yk, yhk, zhk, tensk = sample_rows(Q, 0)
    Wd, Wz = rng.standard_normal((Q, 19)).astype(np.float32), rng.standard_normal((Q, 6)).astype(np.float32)
    out["odep_Wd"], out["odep_Wz"] = Wd, Wz
    for name, sizes, act, hist in [NN_VARIANTS[0], NN_VARIANTS[1], NN_VARIANTS[5]]:
        kk = [v[0] for v in NN_VARIANTS].index(name)
        mlp = orc.make_mlp(sizes, act, seed=100 + kk, history=hist)
        mlp.weights = [w * 3 for w in mlp.weights]
        rob = ref_torch.CosseratRodTorch("cpu", 64, nn_input_history=hist)
        ref_knode.setup_robot(rob, None)
        rob.nn_models = torch_module_list(mlp)
        for use_nn in (0, 1):
            rob.use_nn = bool(use_nn)
            for prm in rob.nn_models.parameters():
                prm.grad = None
            ins = [torch.tensor(a).float().requires_grad_(True) for a in (yk, yhk, zhk)]
            tfk = (torch.tensor(tensk).float() @ rob.tendon_dirs).detach().requires_grad_(True)
            dys, zz = rob.ODE_parallel(ins[0], ins[1], ins[2], tfk)
            L = (dys * torch.tensor(Wd)).sum() + (zz * torch.tensor(Wz)).sum()
            L.backward()
            tag = f"odep_{name}_{use_nn}"
            out[f"{tag}_L"] = np.array(float(L))
            for nm, tns in zip(("dy", "dyh", "dzh", "dtf"), ins + [tfk]):
                out[f"{tag}_{nm}"] = tns.grad.numpy().copy()
            if use_nn:
                for k, prm in enumerate(rob.nn_models.parameters()):
                    out[f"{tag}_dparam{k}"] = prm.grad.numpy().copy()
    save("tres_grad", **out)


def gen_small():
    """F8: calc_controls and quaternion_to_euler."""
    out = {}
    for kind, arg, dt, T in (("sine", 1.0, 0.05, 40), ("sine", 2.5, 0.005, 25), ("step", 1.0, 0.05, 50),
                             ("random", 3.0, 0.05, 30), ("random", 7.9, 0.05, 10)):
        out[f"ctl_{kind}_{arg}_{dt}_{T}"] = np.array(ref_ctl.calc_controls(kind, arg, dt, T))
    rng = np.random.default_rng(3)
    q = rng.standard_normal((4, 400))
    q[:, :40] = np.array([[1.0], [0.0], [0.0], [0.0]]) + 1e-3 * rng.standard_normal((4, 40))
    # near the asin clamp: w*z - x*y ~ +-0.5
    q[:, 40:60] = np.array([[1.0], [0.0], [0.0], [1.0]]) + 1e-4 * rng.standard_normal((4, 20))
    q[:, 60:80] = np.array([[1.0], [0.0], [0.0], [-1.0]]) + 1e-4 * rng.standard_normal((4, 20))
    out["quat"] = q
    out["euler"] = ref_q2e(torch.tensor(q)).numpy()  # reference casts to float32 internally
    save("small", **out)


def gen_round3():
    """Round 3: (a) getResidualRK4 with midpoint histories that are NOT the linear interpolation of yh, zh
    (cosserat_ode.py:225,233-234 read whatever the caller passes); (b) BASELINE cfg2: rods of the
    ``default_rng(1234)`` draw for B = 256 at N = 100, 50 steps, whose fsolve reported ier == 1 throughout."""
    rng = np.random.default_rng(33)
    out = {}
    for N, mod in [(40, "default"), (100, None)]:
        r = np_robot(mod, N)
        y, z, yp, zp, G0, tens = converged_state(
            r, 6, lambda T: ref_ctl.calc_controls("sine", 1.0, r.del_t, T))
        yh = r.c1 * y + r.c2 * yp
        zh = r.c1 * z + r.c2 * zp
        yh_int = 0.5 * (yh[:, :-1] + yh[:, 1:])
        zh_int = 0.5 * (zh[:, :-1] + zh[:, 1:])
        yh_int = yh_int * (1 + 0.1 * rng.standard_normal(yh_int.shape)) + 1e-3 * rng.standard_normal(yh_int.shape)
        zh_int = zh_int * (1 + 0.1 * rng.standard_normal(zh_int.shape)) + 1e-3 * rng.standard_normal(zh_int.shape)
        r.tendon_tensions = tens
        tag = f"mid_N{N}_{mod}"
        out[f"{tag}_y"], out[f"{tag}_z"], out[f"{tag}_yp"], out[f"{tag}_zp"] = y, z, yp, zp
        out[f"{tag}_yh_int"], out[f"{tag}_zh_int"] = yh_int, zh_int
        out[f"{tag}_tens"] = tens
        Gs = G0[None, :] * (1 + 0.05 * rng.standard_normal((2, 6))) + 1e-3 * rng.standard_normal((2, 6))
        out[f"{tag}_G"] = Gs
        rs, ys, zs = [], [], []
        for G in Gs:
            yy, zz = y.copy(), z.copy()
            with np.errstate(all="ignore"):
                rs.append(r.getResidualRK4(G, yy, zz, yh, yh_int, zh, zh_int))
            ys.append(yy)
            zs.append(zz)
        out[f"{tag}_r"], out[f"{tag}_yout"], out[f"{tag}_zout"] = np.array(rs), np.array(ys), np.array(zs)
    # cfg2
    r = np_robot(None, 100)
    T = 50
    ctl_b = orc.batch_sine_controls(256, T, r.del_t, 1234)
    rods, tips, lasts = [], [], []
    cand = 0
    while len(rods) < 8 and cand < 24:
        tr, ie, _ = run_sim(r, ctl_b[cand])
        ok = bool(np.all(ie == 1)) and bool(np.all(np.isfinite(tr)))
        print(f"  cfg2 candidate rod {cand}: ier all 1 = {ok}")
        if ok:
            rods.append(cand)
            tips.append(tr[:, :3, -1])
            lasts.append(tr[-1, :25])
        cand += 1
    out.update(cfg2_rods=np.array(rods), cfg2_tip=np.array(tips), cfg2_last=np.array(lasts), cfg2_T=np.array(T),
               cfg2_seed=np.array(1234), cfg2_B=np.array(256))
    save("round3", **out)


BC_PARAMS = dict(
    F_tip=np.array([0.05, -0.02, 0.1]), M_tip=np.array([1e-3, 2e-3, -1e-3]),
    p0=np.array([0.01, -0.02, 0.03]),
    h0=1.1 * np.array([0.96, 0.10, -0.15, 0.05]),          # tilted AND not of unit length
    q0=np.array([0.01, -0.005, 0.002]), w0=np.array([0.02, -0.01, 0.03]),   # base moving at a constant twist
    tendon_dirs=np.array([[0.70, 0.68, 0.20], [-0.66, 0.75, -0.10], [-0.72, -0.69, 0.15], [0.60, -0.80, -0.25]]),
)


def bc_robot(N, mod=None):
    """A rod with EVERY boundary / load parameter away from its default (cosserat_ode.py:28-29 F_tip, M_tip; :37-41
    tendon_dirs; :44-47 p0, h0, q0, w0 - they enter :194-195 and :206-207)."""
    r = np_robot(mod, N)
    for k, v in BC_PARAMS.items():
        setattr(r, k, v.copy())
    return r


def gen_bc():
    """Round 4: the reference's full parameter surface.  simulate at N = 20 / 100 with a tip wrench, a tilted non-unit
    base quaternion, p0 != 0, a moving base and asymmetric tendon directions with a z component; the Euler and RK4
    residual rows of the same rod; the same with the residual MLP on."""
    rng = np.random.default_rng(44)
    out = {f"par_{k}": v for k, v in BC_PARAMS.items()}
    for N, T in ((20, 30), (100, 30)):
        r = bc_robot(N)
        ctl = ref_ctl.calc_controls("sine", 1.0, r.del_t, T)
        traj, ier, nfev = run_sim(r, ctl)
        print(f"  bc N={N}: ier all 1 = {bool(np.all(ier == 1))}, mean nfev {nfev.mean():.1f}")
        tag = f"sim_N{N}"
        out[f"{tag}_ctl"], out[f"{tag}_ier"] = np.array(ctl), ier
        out[f"{tag}_tip"], out[f"{tag}_last"] = traj[:, :3, -1], traj[-1]
        if N == 20:
            out[f"{tag}_traj"] = traj[:, :25]
        else:
            out[f"{tag}_every10"] = traj[::10, :25]
    # a step input on top (the acceptance / fallback ladders of the persistent kernels see a jump)
    r = bc_robot(40)
    ctl = ref_ctl.calc_controls("step", 1.0, r.del_t, 36)
    traj, ier, _ = run_sim(r, ctl)
    print(f"  bc step N=40: ier all 1 = {bool(np.all(ier == 1))}")
    out["step_N40_ctl"], out["step_N40_ier"], out["step_N40_traj"] = np.array(ctl), ier, traj[:, :25]
    # residual rows
    for N in (20, 100):
        r = bc_robot(N)
        y, z, yp, zp, G0, tens = converged_state(
            r, 6, lambda T: ref_ctl.calc_controls("sine", 1.0, r.del_t, T))
        yh = r.c1 * y + r.c2 * yp
        zh = r.c1 * z + r.c2 * zp
        yh_int = 0.5 * (yh[:, :-1] + yh[:, 1:])
        zh_int = 0.5 * (zh[:, :-1] + zh[:, 1:])
        r.tendon_tensions = tens
        tag = f"res_N{N}"
        out[f"{tag}_y"], out[f"{tag}_z"], out[f"{tag}_yp"], out[f"{tag}_zp"] = y, z, yp, zp
        out[f"{tag}_tens"] = tens
        Gs = G0[None, :] * (1 + 0.05 * rng.standard_normal((3, 6))) + 1e-3 * rng.standard_normal((3, 6))
        out[f"{tag}_G"] = Gs
        for scheme, fn in (("euler", r.getResidualEuler), ("rk4", r.getResidualRK4)):
            rs, ys, zs = [], [], []
            for G in Gs:
                yy, zz = y.copy(), z.copy()
                with np.errstate(all="ignore"):
                    rs.append(fn(G, yy, zz, yh, yh_int, zh, zh_int))
                ys.append(yy)
                zs.append(zz)
            out[f"{tag}_{scheme}_r"] = np.array(rs)
            out[f"{tag}_{scheme}_y"] = np.array(ys)
            out[f"{tag}_{scheme}_z"] = np.array(zs)
    # MLP on: the two-hidden-layer network of BASELINE cfg3 and the single-hidden-layer one of the reference's default shape
    for name, sizes, N, T in (("elu6464", [28, 64, 64, 25], 24, 20), ("elu64", [28, 64, 25], 20, 20)):
        mlp = orc.make_mlp(sizes, "elu", seed=5)
        r = bc_robot(N)
        inject_nn(r, mlp)
        ctl = ref_ctl.calc_controls("sine", 1.5, r.del_t, T)
        traj, ier, _ = run_sim(r, ctl)
        print(f"  bc nn {name} N={N}: ier all 1 = {bool(np.all(ier == 1))}")
        out[f"nn_{name}_ctl"], out[f"nn_{name}_ier"], out[f"nn_{name}_traj"] = np.array(ctl), ier, traj[:, :25]
        out[f"nn_{name}_N"] = np.array(N)
        out.update(mlp_arrays(f"mlp_{name}", mlp))
    save("bc", **out)


ALL = {
    "ode_kat": gen_ode_kat, "ode_torch_kat": gen_ode_torch_kat, "residual_kat": gen_residual_kat,
    "sim_cfg1": gen_sim_cfg1, "sim_n100": gen_sim_n100, "sim_n400": gen_sim_n400, "sim_misc": gen_sim_misc,
    "sim_nn": gen_sim_nn, "sim_more": gen_sim_more, "train_step": gen_train_step, "small": gen_small,
    "checkpoint": gen_checkpoint, "estimate_state": gen_estimate_state, "tres_grad": gen_tres_grad,
    "round3": gen_round3, "bc": gen_bc,
}

if __name__ == "__main__":
    names = sys.argv[1:] or list(ALL)
    for n in names:
        t0 = time.time()
        print(f"[{n}]")
        ALL[n]()
        print(f"  {time.time()-t0:.1f} s")
