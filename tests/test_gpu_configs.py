"""GPU parity tests (``-m gpu``) of the BASELINE.json configurations at their full sizes and of the paths round 1
left without a reference fixture:

  * the reference's DEFAULT network 28 -> 512 -> 25 inside ``simulate`` (sim_more fixture; chunked hidden layer of
    the matrix-core evaluator; the untrained initialisation needs the damped Newton fallback),
  * ``simulate(use_fsolve=False)`` (knode.py:91-94) against the reference's L-BFGS-B run,
  * ``CosseratRodTorch.getResidualEuler(G)`` (cosserat_ode_torch.py:325-367),
  * cfg5: N = 400 under ``calc_controls('sine', P)``, fp64 and fp32, Newton tolerance sweep, B = 512 properties,
  * cfg3: MLP-on ``simulate`` at B = 1024, N = 100 (two rods against the oracle) and the fused training step at
    Q = 193 536 rows against an fp64 torch restatement.

Everything goes through the C ABI; which step kernel ran is asserted (tests/gpu_helpers.py)."""
import numpy as np
import pytest

from conftest import load_golden, rel_l2
from gpu_helpers import MODES, assert_path, inject, make_robot, require_path, set_mode_env

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(params=MODES)
def shooting_mode(request, monkeypatch):
    set_mode_env(monkeypatch, request.param)
    return request.param


def _good_entries(ier):
    """Trajectory entries backed by converged reference solves: solve k produces entry k + 1."""
    return len(ier) - 1 if np.all(ier == 1) else int(np.argmax(ier != 1))


# ---------------------------------------------------------------------------
# reference default network inside simulate
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["elu512", "elu512n24"])
def test_simulate_default_network(torch_cuda, shooting_mode, name):
    import cosserat_oracle as orc
    from knode import simulate
    g = load_golden("sim_more")
    N = int(g[f"{name}_N"])
    mlp = orc.mlp_from_arrays(g, f"mlp_{name}")
    want = require_path(shooting_mode, N, mlp)
    r = make_robot(None, N)
    inject(r, mlp)
    good = _good_entries(g[f"{name}_ier"])
    ref = g[f"{name}_traj"][: good + 1]
    traj = simulate(r, g[f"{name}_ctl"][: good + 1])
    assert_path(r, want)
    assert traj.shape[0] == good + 1
    # the reference stops its own solve at xtol = 1.5e-8; the untrained wide network ("elu512") amplifies what that
    # leaves from step to step (the three kernels land 4e-8 .. 2.5e-7 from it), the scaled one does not
    hard = name == "elu512"
    assert max(rel_l2(traj[t, :25], ref[t]) for t in range(good + 1)) < (1e-6 if hard else 1e-7)
    assert rel_l2(traj[:, :3, -1], ref[:, :3, -1]) < (1e-6 if hard else 1e-8)


# ---------------------------------------------------------------------------
# use_fsolve = False
# ---------------------------------------------------------------------------
def test_simulate_lbfgs_branch(torch_cuda, shooting_mode):
    """knode.py:91-94 minimises the sum of squared residuals with L-BFGS-B; its minimiser is the root the Newton
    solve finds.  The reference's two branches differ by 1.2e-5 (all rows) / 2.9e-6 (tip) on this input - what
    L-BFGS-B's stopping rule leaves - and ours sits on the root side of that gap."""
    from knode import simulate
    g = load_golden("sim_more")
    want = require_path(shooting_mode, 10)
    r = make_robot(None, 10, use_fsolve=False)
    traj = simulate(r, g["lbfgs_ctl"])
    assert_path(r, want)
    ref = g["lbfgs_traj"]
    assert traj.shape == ref.shape
    assert rel_l2(traj[:, :3, -1], ref[:, :3, -1]) < 1e-5
    assert rel_l2(traj[:, :25], ref[:, :25]) < 5e-5
    assert rel_l2(traj[:, 25:], ref[:, 25:]) < 5e-5
    # scalar residual of the class in this mode (cosserat_ode.py:212-213)
    y, z = traj[3, :19].copy(), traj[3, 19:25].copy()
    r.tendon_tensions = g["lbfgs_ctl"][3]
    val = r.getResidualEuler(traj[4, 7:13, 0], y, z, traj[4, 25:44], None, traj[4, 44:50], None)
    assert np.ndim(val) == 0 and 0 <= val < 1e-8


# ---------------------------------------------------------------------------
# a14: differentiable full sweep of the torch twin
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("use_nn", [0, 1])
def test_torch_full_sweep(torch_cuda, use_nn):
    torch = torch_cuda
    from cosserat_ode_torch import CosseratRodTorch
    from knode import setup_robot
    g = load_golden("sim_more")
    rob = CosseratRodTorch(DEV, 64)
    setup_robot(rob, None)
    with torch.no_grad():
        rob.nn_models[0].weight.copy_(torch.tensor(g["mlp_tres_W0"]))
        rob.nn_models[0].bias.copy_(torch.tensor(g["mlp_tres_b0"]))
        rob.nn_models[2].weight.copy_(torch.tensor(g["mlp_tres_W1"]))
        rob.nn_models[2].bias.copy_(torch.tensor(g["mlp_tres_b1"]))
    rob.use_nn = bool(use_nn)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    y, z, yp, zp = t(g["tres_y"]), t(g["tres_z"]), t(g["tres_yp"]), t(g["tres_zp"])
    for k, G in enumerate(g["tres_G"]):
        rob.y, rob.z = y.clone(), z.clone()
        rob.tendon_tensions = t(g["tres_tens"])
        rob.residualArgs["yh"] = rob.c1 * y + rob.c2 * yp
        rob.residualArgs["zh"] = rob.c1 * z + rob.c2 * zp
        total, full = rob.getResidualEuler(t(G))
        ref_val = float(g[f"tres_val_{use_nn}"][k])
        assert full.shape == (25, 10)
        assert abs(float(total.detach()) - ref_val) < 1e-4 * ref_val
        assert full.requires_grad == bool(use_nn)  # like the reference's graph: reaches the MLP parameters
        assert rel_l2(full.detach().cpu().numpy(), g[f"tres_full_{use_nn}"][k]) < 2e-6
        assert rel_l2(rob.y.cpu().numpy(), g[f"tres_yafter_{use_nn}"][k]) < 2e-6


@pytest.mark.parametrize("use_nn", [0, 1])
def test_torch_full_sweep_autograd(torch_cuda, use_nn):
    """a14 with autograd: L = total_residual + sum(full_rod * Wgt); dL/dG and dL/d(every MLP parameter) against the
    reference's autograd through cosserat_ode_torch.py:325-367 (tests/golden/make_golden.py gen_tres_grad; fp32 graph
    there, adjoint sweep with fp64 forward-difference Jacobians here)."""
    torch = torch_cuda
    from cosserat_ode_torch import CosseratRodTorch
    from knode import setup_robot
    g, gg = load_golden("sim_more"), load_golden("tres_grad")
    rob = CosseratRodTorch(DEV, 64)
    setup_robot(rob, None)
    with torch.no_grad():
        rob.nn_models[0].weight.copy_(torch.tensor(g["mlp_tres_W0"]))
        rob.nn_models[0].bias.copy_(torch.tensor(g["mlp_tres_b0"]))
        rob.nn_models[2].weight.copy_(torch.tensor(g["mlp_tres_W1"]))
        rob.nn_models[2].bias.copy_(torch.tensor(g["mlp_tres_b1"]))
    rob.use_nn = bool(use_nn)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    y, z, yp, zp = t(g["tres_y"]), t(g["tres_z"]), t(g["tres_yp"]), t(g["tres_zp"])
    rob.y, rob.z = y.clone(), z.clone()
    rob.tendon_tensions = t(g["tres_tens"])
    rob.residualArgs["yh"] = rob.c1 * y + rob.c2 * yp
    rob.residualArgs["zh"] = rob.c1 * z + rob.c2 * zp
    G = t(g["tres_G"][1]).requires_grad_(True)
    total, full = rob.getResidualEuler(G)
    L = total + (full * t(gg["Wgt"])).sum()
    L.backward()
    assert abs(float(L.detach()) - float(gg[f"L_{use_nn}"])) < 2e-5 * abs(float(gg[f"L_{use_nn}"]))
    assert rel_l2(G.grad.cpu().numpy(), gg[f"dG_{use_nn}"]) < 1e-4
    if use_nn:
        for k, prm in enumerate(rob.nn_models.parameters()):
            assert rel_l2(prm.grad.cpu().numpy(), gg[f"dparam{k}"]) < 1e-3, k
    else:
        assert all(prm.grad is None for prm in rob.nn_models.parameters())
    # the derivative kernels behind the backward passes (kr_ode_jacobian_batch, kr_ode_vjp_batch; fp64 forward mode on the
    # device): the uncut Jacobian against central differences of the HIP ODE kernel itself, the VJP against J^T g and -
    # for the history and tendon-force inputs - against central differences too
    h = rob._native()
    f64 = torch.float64
    yq = y.t().to(f64).contiguous()
    yhq = (rob.c1 * y + rob.c2 * yp).t().to(f64).contiguous()
    zhq = (rob.c1 * z + rob.c2 * zp).t().to(f64).contiguous()
    tf = (t(g["tres_tens"]).to(f64).reshape(1, 4) @ rob.tendon_dirs.to(f64).reshape(4, 3)).expand(10, 3).contiguous()
    F = lambda a, b_, c_, d_: torch.cat(h.ode_batch(a.contiguous(), b_.contiguous(), c_.contiguous(), d_.contiguous(),
                                                   use_nn=False), dim=1)
    J = h.ode_jacobian(yq, yhq, zhq, tf, cut=False)
    assert J.shape == (10, 25, 19)

    def central(which, n):
        cols = []
        args = [yq, yhq, zhq, tf]
        for i in range(n):
            st = 1e-6 * torch.clamp(args[which][:, i].abs(), min=1.0)
            ap = [a.clone() for a in args]
            am = [a.clone() for a in args]
            ap[which][:, i] += st
            am[which][:, i] -= st
            cols.append((F(*ap) - F(*am)) / (2 * st)[:, None])
        return torch.stack(cols, dim=2)   # [Q, 25, n]

    assert rel_l2(J.cpu().numpy(), central(0, 19).cpu().numpy()) < 1e-7
    gen = torch.Generator(device=DEV)
    gen.manual_seed(3)
    gd = torch.randn((10, 25), dtype=f64, device=DEV, generator=gen)
    vy, vyh, vzh, vtf = h.ode_vjp(yq, yhq, zhq, tf, gd[:, :19].contiguous(), gd[:, 19:].contiguous(), cut=False)
    assert rel_l2(vy.cpu().numpy(), torch.einsum("qod,qo->qd", J, gd).cpu().numpy()) < 1e-12
    for got, which, n in ((vyh, 1, 19), (vzh, 2, 6), (vtf, 3, 3)):
        want = torch.einsum("qod,qo->qd", central(which, n), gd)
        assert rel_l2(got.cpu().numpy(), want.cpu().numpy()) < 1e-7
    # cut graph (what the reference's ODE builds): differs from the complete derivative exactly in the h columns
    Jc = h.ode_jacobian(yq, yhq, zhq, tf, cut=True)
    other = [c for c in range(19) if c not in (3, 4, 5, 6)]
    assert torch.equal(Jc[:, :, other][:, [r for r in range(25) if r not in (3, 4, 5, 6)]],
                       J[:, :, other][:, [r for r in range(25) if r not in (3, 4, 5, 6)]])
    assert rel_l2(Jc[:, :, 3:7].cpu().numpy(), J[:, :, 3:7].cpu().numpy()) > 1e-3
    # entries that need no gradient are not written
    only = h.ode_vjp(yq, yhq, zhq, tf, gd[:, :19].contiguous(), gd[:, 19:].contiguous(), need=(False, False, True, False))
    assert only[0] is None and only[1] is None and only[3] is None and torch.equal(only[2], vzh)
    # exact_sweep_gradient: the gradient of the function itself, against forward differences of the fp32 sweep (coarse)
    rob.exact_sweep_gradient = True
    G2 = t(g["tres_G"][1]).requires_grad_(True)
    rob.y, rob.z = y.clone(), z.clone()
    tot2, _ = rob.getResidualEuler(G2)
    tot2.backward()
    eps = 1e-3
    fd = []
    for k in range(6):
        Gp = t(g["tres_G"][1]).double()
        Gp[k] += eps
        rob.y, rob.z = y.clone(), z.clone()
        with torch.no_grad():
            tp, _ = rob.getResidualEuler(Gp.float())
        fd.append((float(tp) - float(tot2.detach())) / eps)
    assert rel_l2(G2.grad.cpu().numpy(), np.array(fd)) < 2e-2


@pytest.mark.parametrize("name,hist", [("elu64", False), ("hist64", True), ("elu6464", False)])
@pytest.mark.parametrize("use_nn", [0, 1])
def test_ode_parallel_input_gradients(torch_cuda, name, hist, use_nn):
    """ODE_parallel with autograd into its inputs (cosserat_ode_torch.py:217-322): L = sum(dys * Wd) + sum(z * Wz),
    dL/d(ys, yhs, zhs, tendon forces) and dL/d(MLP parameters) against the reference's autograd (tres_grad.npz)."""
    torch = torch_cuda
    import torch.nn as nn
    from cosserat_ode_torch import CosseratRodTorch
    from knode import setup_robot
    k, gg = load_golden("ode_kat"), load_golden("tres_grad")
    rob = CosseratRodTorch(DEV, 64, nn_input_history=hist)
    setup_robot(rob, None)
    mods, i = [], 0
    while f"mlp_{name}_W{i}" in k.files:
        W, b = k[f"mlp_{name}_W{i}"], k[f"mlp_{name}_b{i}"]
        lin = nn.Linear(W.shape[1], W.shape[0])
        with torch.no_grad():
            lin.weight.copy_(torch.tensor(W))
            lin.bias.copy_(torch.tensor(b))
        mods.append(lin)
        if f"mlp_{name}_W{i + 1}" in k.files:
            mods.append(nn.ELU())
        i += 1
    rob.nn_models = nn.ModuleList(mods).to(DEV)
    rob.use_nn = bool(use_nn)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    ins = [t(k[n]).requires_grad_(True) for n in ("y", "yh", "zh")]
    tf = (t(k["tensions"]) @ rob.tendon_dirs.float().reshape(4, 3)).detach().requires_grad_(True)
    dys, z = rob.ODE_parallel(ins[0], ins[1], ins[2], tf)
    L = (dys * t(gg["odep_Wd"])).sum() + (z * t(gg["odep_Wz"])).sum()
    L.backward()
    tag = f"odep_{name}_{use_nn}"
    assert abs(float(L.detach()) - float(gg[f"{tag}_L"])) < 1e-5 * abs(float(gg[f"{tag}_L"]))
    for nm, tns in zip(("dy", "dyh", "dzh", "dtf"), ins + [tf]):
        assert rel_l2(tns.grad.cpu().numpy(), gg[f"{tag}_{nm}"]) < 2e-5, nm
    if use_nn:
        for j, prm in enumerate(rob.nn_models.parameters()):
            assert rel_l2(prm.grad.cpu().numpy(), gg[f"{tag}_dparam{j}"]) < 2e-4, j
    # without gradients requested the plain kernel path runs and gives the same values
    with torch.no_grad():
        d2, z2 = rob.ODE_parallel(*[x.detach() for x in ins], tf.detach())
    assert torch.equal(d2, dys.detach()) and torch.equal(z2, z.detach())


# ---------------------------------------------------------------------------
# cfg5: N = 400
# ---------------------------------------------------------------------------
def _n400_case(P):
    if P == "1_0":
        g = load_golden("sim_n400")
        return g["ctl"], g["tip"], g["last"], g["ier"]
    g = load_golden("sim_more")
    return g[f"n400_P{P}_ctl"], g[f"n400_P{P}_tip"], g[f"n400_P{P}_last"], g[f"n400_P{P}_ier"]


@pytest.mark.parametrize("P", ["0_5", "1_0", "2_0", "3_0"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_cfg5_n400_vs_reference(torch_cuda, shooting_mode, P, dtype):
    """calc_controls('sine', P), N = 400: tip path against the reference, fp64 <= 1e-8, fp32 inside the 1e-5 contract."""
    from knode import simulate_batch
    ctl, tip, last, ier = _n400_case(P)
    assert np.all(ier == 1)
    want = require_path(shooting_mode, 400)
    r = make_robot(None, 400)
    T = len(tip) - 1  # entry 0 is the initial tip
    out = simulate_batch(r, ctl[None, :T], dtype=dtype)
    assert_path(r, want)
    assert np.all(out["status"] == 0)
    got = np.concatenate([out["traj"][0, :1, :3, -1], out["tip"][0]])
    assert rel_l2(got, tip) < (1e-8 if dtype == "f64" else 1e-5)
    assert rel_l2(out["traj"][0, T], last) < (1e-7 if dtype == "f64" else 2e-5)


@pytest.mark.parametrize("dtype,tol", [("f64", 1e-6), ("f64", 1e-8), ("f64", 1e-10), ("f64", 1e-12),
                                       ("f32", 1e-3), ("f32", 1e-4), ("f32", 1e-5), ("f32", 1e-6)])
def test_cfg5_tolerance_sweep(torch_cuda, shooting_mode, dtype, tol):
    """BASELINE cfg5 "fp64 vs fp32 tolerance sweep": the Newton stopping tolerance against the tip error to the
    reference (sim_n400, 12 steps).  The solver stops on the UPDATE, which overestimates the error left behind
    (quadratic convergence), so even the loosest setting stays far inside the contract; fp32 bottoms out at its
    rounding level."""
    from knode import simulate_batch
    g = load_golden("sim_n400")
    want = require_path(shooting_mode, 400)
    r = make_robot(None, 400)
    T = len(g["tip"]) - 1
    out = simulate_batch(r, g["ctl"][None, :T], dtype=dtype, tol=tol, maxit=60)
    assert_path(r, want)
    assert np.all(out["status"] == 0)
    err = rel_l2(np.concatenate([out["traj"][0, :1, :3, -1], out["tip"][0]]), g["tip"])
    bound = max(1e-8, 10 * tol) if dtype == "f64" else max(5e-6, 10 * tol)
    assert err < min(bound, 1e-5 if tol <= 1e-4 else 1e-2), (dtype, tol, err)


def test_cfg5_full_size_properties(torch_cuda, shooting_mode):
    """B = 512 rods x N = 400 (the cfg5 batch), fp64 and fp32: every step converges, the stored state is a root of
    the shooting residual, a rod's result does not depend on the batch around it, fp32 stays within 1e-5 of fp64 at
    the tip, and two rods match the C oracle."""
    torch = torch_cuda
    import cosserat_oracle as orc
    import cosserat_oracle_c as oc
    want = require_path(shooting_mode, 400)
    r = make_robot(None, 400)
    h = r._native()
    B, T = 512, 5
    Ps = np.array([0.5, 1.0, 2.0, 3.0])
    ctl = np.stack([np.array(orc.calc_controls("sine", float(Ps[b % 4]), r.del_t, T)) * (1.0 + 0.02 * (b // 4) / (B // 4))
                    for b in range(B)])
    tips = {}
    for dt in (torch.float64, torch.float32):
        ctl_t = torch.as_tensor(ctl, device=DEV).to(dt).contiguous()
        states = h.new_state(B, dt, n_slots=T + 1)
        h.init_straight(states[0])
        G = torch.zeros((B, 6), dtype=dt, device=DEV)
        status = torch.full((B, T), -1, dtype=torch.int32, device=DEV)
        tip = torch.empty((B, T, 3), dtype=dt, device=DEV)
        h.simulate(ctl_t, states, G, tip=tip, status=status)
        torch.cuda.synchronize()
        assert_path(h, want)
        assert int((status != 0).sum()) == 0
        tips[dt] = tip.double().cpu().numpy()
        nxt = h.new_state(B, dt)
        res = h.residual(G, states[T - 2], states[T - 1], nxt, ctl_t[:, T - 1].contiguous())
        tol_r = 1e-8 if dt == torch.float64 else 2e-3
        assert float(res.abs().max()) < tol_r * max(1.0, float(G.abs().max()))
        st2 = h.new_state(3, dt, n_slots=T + 1)
        h.init_straight(st2[0])
        G2 = torch.zeros((3, 6), dtype=dt, device=DEV)
        h.simulate(ctl_t[:3].contiguous(), st2, G2)
        assert torch.equal(st2[T], states[T][:3])
        assert float(states[T][..., 25:].abs().max()) == 0.0
    for b in range(B):
        e = rel_l2(tips[torch.float32][b], tips[torch.float64][b])
        assert e < 1e-5, (b, e)
    for b in (1, 510):
        tip_c, _, bad = oc.simulate(orc.params_for(None, 400), ctl[b])
        assert bad == 0 and rel_l2(tips[torch.float64][b], tip_c) < 1e-8


@pytest.mark.parametrize("mode", ["persistent", "overlap"])
def test_fp32_large_batch_two_waves_per_simd(torch_cuda, monkeypatch, mode):
    """fp32 batches beyond one rod per SIMD run the persistent kernels' instantiation that keeps two wavefronts on a
    SIMD (kr_ms_impl.hpp / kr_mso_impl.hpp, OCC = 2): every step converges, rods equal the same rods simulated in a
    small batch (the one-wavefront-per-SIMD instantiation) to fp32 rounding, tips inside the 1e-5 contract against the
    fp64 oracle.  The batch size is ragged on purpose (last workgroup partly empty)."""
    torch = torch_cuda
    import cosserat_oracle as orc
    set_mode_env(monkeypatch, mode)
    r = make_robot(None, 100)
    h = r._native()
    dt = torch.float32
    B, T = 2050, 30
    ctl = orc.batch_sine_controls(B, T, r.del_t, 321)
    ctl_t = torch.as_tensor(ctl, device=DEV).to(dt).contiguous()
    st = h.new_state(B, dt, n_slots=T + 1)
    h.init_straight(st[0])
    status = torch.full((B, T), -1, dtype=torch.int32, device=DEV)
    tip = torch.empty((B, T, 3), dtype=dt, device=DEV)
    h.simulate(ctl_t, st, torch.zeros((B, 6), dtype=dt, device=DEV), status=status, tip=tip)
    assert_path(h, 3 if mode == "overlap" else 2)
    assert int((status != 0).sum()) == 0
    pick = [0, 1, 1023, 1024, 2047, 2049]
    small = h.new_state(len(pick), dt, n_slots=T + 1)
    h.init_straight(small[0])
    h.simulate(ctl_t[pick].contiguous(), small, torch.zeros((len(pick), 6), dtype=dt, device=DEV))
    assert float((small[T] - st[T][pick]).abs().max()) < 2e-5 * float(small[T].abs().max())
    D = orc.params_for(None, 100).derived()
    for b in (1, 2047):
        ref = orc.simulate(D, np.vstack([ctl[b], ctl[b][-1:]]), solver="newton")[1:, :3, -1]
        assert rel_l2(tip[b].double().cpu().numpy(), ref) < 1e-5


# ---------------------------------------------------------------------------
# cfg3 at full size
# ---------------------------------------------------------------------------
def _cfg3_mlp():
    import cosserat_oracle as orc
    return orc.make_mlp([28, 64, 64, 25], "elu", seed=7)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_cfg3_full_size_nn_simulate(torch_cuda, shooting_mode, dtype):
    """B = 1024 rods, N = 100, MLP 28 -> 64 -> 64 -> 25 inside every sweep (SURVEY 8d cfg3 inputs, rng 1235):
    all steps converge, two rods of the batch against the oracle (fp64 <= 1e-7; fp32 tip <= 1e-5)."""
    torch = torch_cuda
    import cosserat_oracle as orc
    from knode import simulate_batch
    mlp = _cfg3_mlp()
    want = require_path(shooting_mode, 100, mlp)
    r = make_robot(None, 100)
    inject(r, mlp)
    B, T = 1024, 8
    ctl = orc.batch_sine_controls(B, T, r.del_t, 1235)
    out = simulate_batch(r, ctl, dtype=dtype, tip_only=True)
    assert_path(r, want)
    assert np.all(out["status"] == 0)
    D = orc.params_for(None, 100).derived()
    for b in (3, 1000):
        ref = orc.simulate(D, np.vstack([ctl[b], ctl[b][-1:]]), mlp=mlp, solver="newton")[1:, :3, -1]
        assert rel_l2(out["tip"][b], ref) < (1e-7 if dtype == "f64" else 1e-5)
    # the network must matter at this scale, otherwise the check pins nothing
    plain = orc.simulate(D, np.vstack([ctl[3], ctl[3][-1:]]), solver="newton")[1:, :3, -1]
    assert rel_l2(out["tip"][3], plain) > 1e-4


def test_cfg3_full_size_training_step(torch_cuda):
    """The fused training step at the cfg3 size: 1024 trajectories x 63 window steps x 3 key points = 193 536 rows,
    28 -> 64 -> 64 -> 25.  Inputs of the MLP and the parameter-free part of the prediction are spot-checked against
    the oracle; loss and every parameter gradient against an fp64 torch restatement of physics_train.py:250-267 on
    the same rows; the Adam + clamp update against torch.optim.Adam."""
    torch = torch_cuda
    import torch.nn as nn
    import cosserat_oracle as orc
    from cosserat_ode_torch import CosseratRodTorch
    from knode import setup_robot, simulate_batch
    from krod_train import KnodeTrainer
    from Utils.transformations import quaternion_to_euler
    M, T, N = 1024, 64, 100
    kp = [22, 67, 99]  # round(N * [2, 6, 9] / 9), SURVEY 8d
    rr = make_robot(None, N)
    ctl = orc.batch_sine_controls(M, T, rr.del_t, 1236)
    o = simulate_batch(rr, ctl, dtype="f32")
    assert np.all(o["status"] == 0)
    traj = torch.as_tensor(o["traj"][:, :T], device=DEV).float().contiguous()
    controls = torch.as_tensor(ctl, device=DEV).float().contiguous()
    rob = CosseratRodTorch(DEV, 64)
    setup_robot(rob, "damping")
    rob.N = N
    rob.compute_intermediate_terms()
    torch.manual_seed(5)
    mods = [nn.Linear(28, 64), nn.ELU(), nn.Linear(64, 64), nn.ELU(), nn.Linear(64, 25)]
    for m in mods:
        if isinstance(m, nn.Linear):
            rob.non_negative_normal_init(m, 0.01, 0.01)
            nn.init.normal_(m.bias, 0.0, 0.01)
    rob.nn_models = nn.ModuleList(mods).to(DEV)
    tr = KnodeTrainer(rob, traj, controls, kp)
    assert tr.Q == 193536
    # (1) rows: x = [y, z, tf] at column key-1 of the teacher-forced next state, base = y + ds * physics (z: physics)
    D = orc.setup_params("damping", N).derived()
    x = tr.x.cpu().numpy()
    base = tr.base.cpu().numpy()
    tj = traj.cpu().numpy().astype(np.float64)
    for q in (0, 77777, 193535):
        s, k = divmod(q, 3)
        m, t = divmod(s, T - 1)
        col = kp[k] - 1
        G = tj[m, t + 1]
        y, z = tj[m, t, :19], tj[m, t, 19:]
        yp, zp = (y, z) if t == 0 else (tj[m, t - 1, :19], tj[m, t - 1, 19:])
        yh, zh = D.c1 * y + D.c2 * yp, D.c1 * z + D.c2 * zp
        tf = orc.tendon_force(D, ctl[m, t].astype(np.float32).astype(np.float64))
        ys, zz = orc.ode(D, G[:19, col], yh[:, col], zh[:, col], tf)
        want_x = np.concatenate([G[:19, col], zz, tf])
        want_b = np.concatenate([G[:19, col] + D.ds * ys, zz])
        assert np.allclose(x[q, :28], want_x, rtol=2e-4, atol=2e-5 * np.abs(want_x).max()), q
        assert np.allclose(base[q], want_b, rtol=2e-4, atol=2e-5 * np.abs(want_b).max()), q
    # (2) loss and gradients against fp64 torch on the same rows
    loss = tr.loss_and_grads()
    torch.cuda.synchronize()
    got_loss = float(loss.item())
    got_grads = [p.grad.detach().clone() for p in rob.nn_models.parameters()]
    ref_net = nn.Sequential(*[nn.Linear(m.in_features, m.out_features) if isinstance(m, nn.Linear) else nn.ELU()
                              for m in mods]).to(DEV).double()
    with torch.no_grad():
        for a, b in zip(ref_net.parameters(), rob.nn_models.parameters()):
            a.copy_(b.double())
    out = ref_net(tr.x[:, :28].double())
    pred = tr.base.double() + torch.cat([float(rob.ds) * out[:, :19], out[:, 19:]], dim=1)
    tgt = tr.target_rows[: tr.Q].double()
    K, steps = 3, T - 1
    e_pred = quaternion_to_euler(pred[:, 3:7].t().float()).double()  # the reference's loss runs this part in fp32
    e_tgt = quaternion_to_euler(tgt[:, 3:7].t().float()).double()
    # quaternion_to_euler casts to fp32, which cuts the fp64 graph's precision but not its gradient path
    total = (((pred[:, :3] - tgt[:, :3]) ** 2).sum() / (3 * K) + ((pred[:, 7:19] - tgt[:, 7:19]) ** 2).sum() / (12 * K)
             + ((e_pred - e_tgt) ** 2).sum() / (3 * K) + ((pred[:, 19:] - tgt[:, 19:]) ** 2).sum() / (6 * K)) / steps
    total.backward()
    assert abs(got_loss - float(total)) < 1e-4 * abs(float(total))
    for a, p in zip(got_grads, ref_net.parameters()):
        assert rel_l2(a.cpu().numpy(), p.grad.cpu().numpy()) < 2e-4
    # (3) update: Adam(lr 1e-2) + clamp of every weight matrix
    before = [p.detach().clone() for p in rob.nn_models.parameters()]
    opt = torch.optim.Adam([nn.Parameter(b.clone()) for b in before], lr=1e-2)
    for q_, g_ in zip(opt.param_groups[0]["params"], got_grads):
        q_.grad = g_.clone()
    opt.step()
    tr.apply_update()
    for k, (p, q_) in enumerate(zip(rob.nn_models.parameters(), opt.param_groups[0]["params"])):
        want_p = q_.detach().clamp(min=0) if k % 2 == 0 else q_.detach()
        assert float((p.detach() - want_p).abs().max()) < 1e-6
    # (4) the SAME epoch through kr_train_epoch (tr.step(): mlp_fwd3_kernel + loss epilogue, mlp_bwd3_kernel,
    # train_tail_kernel - the kernels bench.py's cfg3_train_epoch leg times) from the same initial weights, against the
    # fp64 torch loss / gradients and torch.optim.Adam above, not against its separate-call sibling
    rob2 = CosseratRodTorch(DEV, 64)
    setup_robot(rob2, "damping")
    rob2.N = N
    rob2.compute_intermediate_terms()
    mods2 = [nn.Linear(28, 64), nn.ELU(), nn.Linear(64, 64), nn.ELU(), nn.Linear(64, 25)]
    rob2.nn_models = nn.ModuleList(mods2).to(DEV)
    with torch.no_grad():
        for a, b in zip(rob2.nn_models.parameters(), before):
            a.copy_(b)
    tr2 = KnodeTrainer(rob2, traj, controls, kp)
    l_epoch = tr2.step()
    assert tr2.fused_epoch
    assert abs(l_epoch - float(total)) < 1e-4 * abs(float(total))
    off = 0
    for k, (p, q_, gref) in enumerate(zip(rob2.nn_models.parameters(), opt.param_groups[0]["params"], ref_net.parameters())):
        want_p = q_.detach().clamp(min=0) if k % 2 == 0 else q_.detach()
        assert float((p.detach() - want_p).abs().max()) < 1e-6, k
        n = p.numel()
        g64 = gref.grad.reshape(-1)
        m = tr2.exp_avg[off:off + n].double()
        assert float((m - 0.1 * g64).norm() / (0.1 * g64).norm()) < 2e-4, k  # exp_avg after step 1 = (1 - beta1) g
        off += n
