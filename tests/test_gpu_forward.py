"""GPU parity tests of the forward path (run with ``-m gpu`` on the MI355X box).

Everything goes through the reference-shaped Python surface
(``cosserat_ode.CosseratRod``, ``knode.setup_robot`` / ``simulate``), i.e. through
the C ABI of libknode_rod.so, and is compared with

  * the golden vectors the reference produced (tests/golden/*.npz), and
  * the CPU oracle (oracle/cosserat_oracle.py) on seeded inputs.

Tolerances: fp64 results are compared at 1e-9..1e-12 relative L2 (the reference
itself only converges its shooting solve to xtol=1.5e-8); the contract of
BASELINE.json is tip-trajectory relative L2 <= 1e-5, which is what the fp32
tests assert.
"""
import numpy as np
import pytest

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu

MODS = ["default", None, "noair", "nsw", "short", "damping", "dampstiff", "lengthstiff", "youngs"]
NN = ["elu64", "hist64", "tanh6464", "softplus6464", "relu6464", "elu6464"]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


from gpu_helpers import MODES, assert_path, expected_path, inject, make_robot, require_path, set_mode_env  # noqa: E402


@pytest.fixture(autouse=True, params=MODES)
def shooting_mode(request, monkeypatch):
    """Every test runs three times: with the single-shooting step kernel (8 rods per wavefront), with
    the multiple-shooting one (1 rod per wavefront, one launch per step) and with its persistent form
    (all steps of kr_simulate_batch in one launch).  Tests that simulate call ``require_path`` (skips the
    parametrisation when the named kernel cannot serve the problem) and ``assert_path`` (what ran)."""
    set_mode_env(monkeypatch, request.param)
    return request.param


def _once(mode):
    """Kernels that take no time step (batched ODE, residual sweeps, MLP rows) do not depend on the step-kernel
    mode: run under one parametrisation only."""
    if mode != "single":
        pytest.skip("independent of the step-kernel mode")


# ---------------------------------------------------------------------------
# K1: batched ODE
# ---------------------------------------------------------------------------
def _ode_rows_gpu(torch, robot, g, dtype):
    dev = "cuda:0"
    h = robot._native()
    tf = g["tensions"] @ robot.tendon_dirs
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev).to(dtype).contiguous()
    dys, z = h.ode_batch(t(g["y"]), t(g["yh"]), t(g["zh"]), t(tf), use_nn=robot._use_nn)
    return torch.cat([dys, z], 1).cpu().numpy()


@pytest.mark.parametrize("mod", MODS)
def test_ode_batch_presets_f64(torch_cuda, shooting_mode, mod):
    _once(shooting_mode)
    g = load_golden("ode_kat")
    got = _ode_rows_gpu(torch_cuda, make_robot(mod, 10), g, torch_cuda.float64)
    ref = g[f"phys_{mod}"]
    assert rel_l2(got, ref) < 1e-13
    assert np.max(np.abs(got - ref) / (np.abs(ref) + 1e-9 * np.abs(ref).max())) < 1e-8


@pytest.mark.parametrize("name", NN)
def test_ode_batch_mlp_f64(torch_cuda, shooting_mode, name):
    _once(shooting_mode)
    import cosserat_oracle as orc
    g = load_golden("ode_kat")
    r = make_robot(None, 10)
    inject(r, orc.mlp_from_arrays(g, f"mlp_{name}"))
    got = _ode_rows_gpu(torch_cuda, r, g, torch_cuda.float64)
    assert rel_l2(got, g[f"nn_{name}"]) < 1e-12


@pytest.mark.parametrize("name", ["elu64", "hist64", "elu6464"])
@pytest.mark.parametrize("use_nn", [0, 1])
def test_ode_batch_f32_vs_torch_twin(torch_cuda, shooting_mode, name, use_nn):
    """fp32 kernel against CosseratRodTorch.ODE_parallel / ODE outputs."""
    _once(shooting_mode)
    import cosserat_oracle as orc
    g = load_golden("ode_kat")
    gt = load_golden("ode_torch_kat")
    r = make_robot(None, 10)
    if use_nn:
        inject(r, orc.mlp_from_arrays(g, f"mlp_{name}"))
    got = _ode_rows_gpu(torch_cuda, r, g, torch_cuda.float32)
    ref = gt[f"par_{name}_{use_nn}"]
    # the two fp32 statements of the reference agree with each other to ~1e-6; ask the same of ours
    scale = np.abs(ref).max(axis=0, keepdims=True)
    assert np.max(np.abs(got - ref) / scale) < 5e-5
    assert rel_l2(got, ref) < 2e-6


def test_ode_single_row_method(torch_cuda, shooting_mode):
    _once(shooting_mode)
    g = load_golden("ode_kat")
    r = make_robot(None, 10)
    i = 7
    ys, z = r.ODE(g["y"][i], g["yh"][i], g["zh"][i], g["tensions"][i] @ r.tendon_dirs)
    assert ys.shape == (19,) and z.shape == (6,)
    assert rel_l2(np.concatenate([ys, z]), g["phys_None"][i]) < 1e-13


def test_ode_ragged_sizes(torch_cuda, shooting_mode):
    """Q not a multiple of the tile, Q = 1, Q = 0."""
    _once(shooting_mode)
    torch = torch_cuda
    g = load_golden("ode_kat")
    r = make_robot(None, 10)
    h = r._native()
    tf = g["tensions"] @ r.tendon_dirs
    full = _ode_rows_gpu(torch, r, g, torch.float64)
    for Q in (1, 47, 0):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a[:Q]), device="cuda:0").contiguous()
        dys, z = h.ode_batch(t(g["y"]), t(g["yh"]), t(g["zh"]), t(tf))
        assert dys.shape == (Q, 19)
        if Q:
            assert np.array_equal(torch.cat([dys, z], 1).cpu().numpy(), full[:Q])
    # many tiles: replicate rows
    reps = 700
    big = lambda a: torch.as_tensor(np.tile(a, (reps, 1)), device="cuda:0").contiguous()
    dys, z = h.ode_batch(big(g["y"]), big(g["yh"]), big(g["zh"]), big(tf))
    out = torch.cat([dys, z], 1).cpu().numpy().reshape(reps, -1, 25)
    assert np.array_equal(out[0], full) and np.array_equal(out[-1], full)


# ---------------------------------------------------------------------------
# shooting residuals
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["N10_None", "N20_None", "N100_None", "N10_default", "N40_default"])
@pytest.mark.parametrize("scheme", ["euler", "rk4"])
def test_residual_methods(torch_cuda, shooting_mode, tag, scheme):
    _once(shooting_mode)
    g = load_golden("residual_kat")
    N = int(tag.split("_")[0][1:])
    mod = tag.split("_")[1]
    r = make_robot("default" if mod == "default" else None, N)
    y0, z0, yp, zp = g[f"{tag}_y"], g[f"{tag}_z"], g[f"{tag}_yp"], g[f"{tag}_zp"]
    yh = r.c1 * y0 + r.c2 * yp
    zh = r.c1 * z0 + r.c2 * zp
    yh_int = 0.5 * (yh[:, :-1] + yh[:, 1:])
    zh_int = 0.5 * (zh[:, :-1] + zh[:, 1:])
    r.tendon_tensions = g[f"{tag}_tens"]
    fn = r.getResidualEuler if scheme == "euler" else r.getResidualRK4
    for k, G in enumerate(g[f"{tag}_G"]):
        ref_r, ref_y, ref_z = g[f"{tag}_{scheme}_r"][k], g[f"{tag}_{scheme}_y"][k], g[f"{tag}_{scheme}_z"][k]
        if not np.all(np.isfinite(ref_r)):
            continue  # the reference itself diverges (RK4 on a coarse grid)
        y, z = y0.copy(), z0.copy()
        res = fn(G, y, z, yh, yh_int, zh, zh_int)
        assert rel_l2(y, ref_y) < 1e-10
        assert rel_l2(z, ref_z) < 1e-10
        assert np.allclose(res, ref_r, rtol=1e-8, atol=1e-10 * np.abs(ref_y[7:13]).max())
        assert np.array_equal(z[:, -1], z0[:, -1])  # never written
    # scalar form when use_fsolve is off (cosserat_ode.py:210-213)
    r.use_fsolve = False
    y, z = y0.copy(), z0.copy()
    val = r.getResidualEuler(g[f"{tag}_G"][0], y, z, yh, yh_int, zh, zh_int)
    assert np.isscalar(val) or np.ndim(val) == 0


# ---------------------------------------------------------------------------
# simulate
# ---------------------------------------------------------------------------
def test_simulate_cfg1(torch_cuda, shooting_mode):
    """BASELINE config 1: single rod, N=20, 200 steps, tensions [6,5,5,6]."""
    from knode import simulate
    g = load_golden("sim_cfg1")
    want = require_path(shooting_mode, 20)
    r = make_robot(None, 20)
    traj = simulate(r, g["ctl"])
    assert_path(r, want)
    assert traj.shape == (200, 50, 20) and traj.dtype == np.float64
    assert rel_l2(traj[:, :3, -1], g["tip"]) < 1e-8
    assert rel_l2(traj[::10, :25], g["every10"]) < 1e-7
    assert rel_l2(traj[-1], g["last"]) < 1e-7


def test_simulate_full50_layout(torch_cuda, shooting_mode):
    from knode import simulate
    g = load_golden("sim_misc")
    want = require_path(shooting_mode, 10)
    r = make_robot(None, 10)
    traj = simulate(r, g["full50_ctl"])
    assert_path(r, want)
    ref = g["full50_traj"]
    assert traj.shape == ref.shape
    assert np.array_equal(traj[0], ref[0])  # initial entry incl. its [y;z;y;z] quirk
    for lo, hi in ((0, 19), (19, 25), (25, 44), (44, 50)):
        assert rel_l2(traj[:, lo:hi], ref[:, lo:hi]) < 1e-8


@pytest.mark.parametrize("mod", MODS[2:] + ["default"])
def test_simulate_presets(torch_cuda, shooting_mode, mod):
    from knode import simulate
    g = load_golden("sim_misc")
    want = require_path(shooting_mode, 10)
    r = make_robot(mod, 10)
    traj = simulate(r, g[f"mod_{mod}_ctl"])
    assert_path(r, want)
    assert rel_l2(traj[:, :25], g[f"mod_{mod}_traj"]) < 1e-8


@pytest.mark.parametrize("kind", ["step", "random"])
def test_simulate_inputs(torch_cuda, shooting_mode, kind):
    from knode import simulate
    g = load_golden("sim_misc")
    want = require_path(shooting_mode, 10)
    r = make_robot(None, 10)
    traj = simulate(r, g[f"{kind}_ctl"])
    assert_path(r, want)
    assert rel_l2(traj[:, :25], g[f"{kind}_traj"]) < 1e-8


def test_simulate_n100_and_batch(torch_cuda, shooting_mode):
    from knode import simulate, simulate_batch
    g = load_golden("sim_n100")
    want = require_path(shooting_mode, 100)
    r = make_robot(None, 100)
    traj = simulate(r, g["ctl"])
    assert_path(r, want)
    assert rel_l2(traj[:, :3, -1], g["tip"]) < 1e-8
    assert rel_l2(traj[::10, :25], g["every10"]) < 1e-7
    # cfg2-style batch: 6 rods with random phase / period, compared rod by rod
    out = simulate_batch(r, g["batch_ctl"])
    assert_path(r, want)
    assert np.all(out["status"] == 0)
    ref = g["batch_tip"]  # [B, T, 3], entry 0 = initial tip, entry t = after step t
    got = out["traj"][:, : ref.shape[1], :3, -1]
    for b in range(ref.shape[0]):
        assert rel_l2(got[b], ref[b]) < 1e-8
    # tip output of the kernel = tip of the stored state
    assert np.array_equal(out["tip"], out["traj"][:, 1:, :3, -1])
    # fp32 arithmetic stays inside the 1e-5 contract
    out32 = simulate_batch(r, g["batch_ctl"], dtype="f32")
    assert_path(r, want)
    assert np.all(out32["status"] == 0)
    for b in range(ref.shape[0]):
        assert rel_l2(out32["traj"][b, : ref.shape[1], :3, -1], ref[b]) < 1e-5


def test_simulate_n400(torch_cuda, shooting_mode):
    """N=400: the single-shooting kernel takes its history from global scratch (8 rods x 400 points do not fit
    the LDS), the multiple-shooting kernel runs one or two rods per workgroup."""
    from knode import simulate
    g = load_golden("sim_n400")
    want = require_path(shooting_mode, 400)
    r = make_robot(None, 400)
    traj = simulate(r, g["ctl"])
    assert_path(r, want)
    assert rel_l2(traj[:, :3, -1], g["tip"]) < 1e-8
    assert rel_l2(traj[-1, :25], g["last"]) < 1e-7


def test_simulate_rk4(torch_cuda, shooting_mode):
    from knode import simulate_batch
    g = load_golden("sim_misc")
    want = require_path(shooting_mode, 40, scheme="rk4")
    r = make_robot(None, 40)
    ctl = g["rk4_ctl"]
    out = simulate_batch(r, ctl[None], scheme="rk4")
    assert_path(r, want)
    ref = g["rk4_traj"]  # [T, 25, N] entries 0..T-1
    assert rel_l2(out["traj"][0, : ref.shape[0]], ref) < 1e-7


@pytest.mark.parametrize("name", ["elu64", "elu6464", "hist64"])
def test_simulate_with_mlp(torch_cuda, shooting_mode, name):
    import cosserat_oracle as orc
    from knode import simulate
    g = load_golden("sim_nn")
    mlp = orc.mlp_from_arrays(g, f"mlp_{name}")
    want = require_path(shooting_mode, int(g[f"{name}_N"]), mlp)
    r = make_robot(None, int(g[f"{name}_N"]))
    inject(r, mlp)
    traj = simulate(r, g[f"{name}_ctl"])
    assert_path(r, want)
    assert rel_l2(traj[:, :25], g[f"{name}_traj"]) < 1e-8


def test_get_nn_output(torch_cuda, shooting_mode):
    _once(shooting_mode)
    import cosserat_oracle as orc
    g = load_golden("ode_kat")
    mlp = orc.mlp_from_arrays(g, "mlp_tanh6464")
    r = make_robot(None, 10)
    inject(r, mlp)
    x = np.random.default_rng(0).standard_normal(28)
    got = r.get_nn_output(x, r.nn_model, r.param_ls)
    assert rel_l2(got, orc.mlp_eval(mlp, x)) < 1e-13


# ---------------------------------------------------------------------------
# full-size, size-independent properties (BASELINE: B=1024, N=100)
# ---------------------------------------------------------------------------
def test_full_size_properties(torch_cuda, shooting_mode):
    torch = torch_cuda
    import cosserat_oracle as orc
    import krod_native as kn
    want = require_path(shooting_mode, 100)
    r = make_robot(None, 100)
    h = r._native()
    B, T = 1024, 6
    ctl = orc.batch_sine_controls(B, T, r.del_t, 1235)
    dev = "cuda:0"
    ctl_t = torch.as_tensor(ctl, device=dev).contiguous()
    states = h.new_state(B, torch.float64, n_slots=T + 1)
    h.init_straight(states[0])
    G = torch.zeros((B, 6), dtype=torch.float64, device=dev)
    status = torch.full((B, T), -1, dtype=torch.int32, device=dev)
    tip = torch.empty((B, T, 3), dtype=torch.float64, device=dev)
    h.simulate(ctl_t, states, G, tip=tip, status=status)
    torch.cuda.synchronize()
    assert_path(h, want)
    assert int((status != 0).sum()) == 0
    # (1) the stored state is a root of the shooting residual: re-sweeping from the returned G
    #     reproduces it bit for bit and leaves a tiny residual
    nxt = h.new_state(B, torch.float64)
    res = h.residual(G, states[T - 2], states[T - 1], nxt, ctl_t[:, T - 1].contiguous())
    assert float(res.abs().max()) < 1e-8
    if shooting_mode == "single":
        assert torch.equal(nxt[..., :25], states[T][..., :25])
    else:  # interface jumps of the sub-intervals are below the solver tolerance, not zero
        assert float((nxt[..., :25] - states[T][..., :25]).abs().max()) < 1e-8
    # (2) a rod's result does not depend on what else is in the batch (first 8 rods alone; one rod alone)
    for nb in (8, 1, 13):
        st2 = h.new_state(nb, torch.float64, n_slots=T + 1)
        h.init_straight(st2[0])
        G2 = torch.zeros((nb, 6), dtype=torch.float64, device=dev)
        h.simulate(ctl_t[:nb].contiguous(), st2, G2)
        assert torch.equal(st2[T], states[T][:nb])
    # (3) oracle spot check on two rods of the big batch
    D = orc.params_for(None, 100).derived()
    for b in (0, 777):
        tr = orc.simulate(D, np.vstack([ctl[b], ctl[b][-1:]]), solver="fsolve")
        assert rel_l2(tip[b].cpu().numpy(), tr[1:, :3, -1]) < 1e-7
    # (4) padding slots stay zero
    assert float(states[T][..., 25:].abs().max()) == 0.0


@pytest.mark.parametrize("kind", ["sine_fast", "step", "random"])
def test_adaptive_predictor_on_rough_inputs(torch_cuda, shooting_mode, kind):
    """The persistent kernel extrapolates the unknowns in time with an order (<= 7) it picks per rod and
    step.  On inputs that are not smooth (a jump; fresh random tensions every step, physics_controls.py:
    22-30) it has to fall back to low orders: every step must still converge, to the same states as the
    reference's plain warm start (predictor 0, one launch per step)."""
    torch = torch_cuda
    if shooting_mode not in ("persistent", "overlap"):
        pytest.skip("compares the persistent kernels with the per-step one itself")
    r = make_robot(None, 40)
    h = r._native()
    B, T = 16, 90
    rng = np.random.default_rng(5)
    i = np.arange(1, T + 1)[None, :, None]
    k = np.arange(4)[None, None, :]
    if kind == "sine_fast":
        per = rng.uniform(0.4, 0.6, size=(B, 1, 1))
        ctl = 6.0 + np.sin(2 * np.pi * i * r.del_t / per + k * np.pi / 2)
    elif kind == "step":
        ctl = np.full((B, T, 4), 5.0)
        jump = rng.uniform(0.5, 2.0, size=(B, 1))
        ctl[:, 30:, 0] += jump
        ctl[:, 30:, 3] += jump
        ctl[:, 60:, 1] += 0.5 * jump
    else:
        ctl = 5.0 + 5.0 * rng.uniform(size=(B, T, 4))
    dev = "cuda:0"
    ctl_t = torch.as_tensor(ctl, device=dev).contiguous()
    outs = []
    for pred, persistent in ((8, 1), (7, 1), (0, 0)):
        h.set_option("predictor", pred)
        h.set_option("persistent", persistent)
        st = h.new_state(B, torch.float64, n_slots=T + 1)
        h.init_straight(st[0])
        G = torch.zeros((B, 6), dtype=torch.float64, device=dev)
        status = torch.full((B, T), -1, dtype=torch.int32, device=dev)
        h.simulate(ctl_t, st, G, status=status)
        torch.cuda.synchronize()
        assert_path(h, (3 if shooting_mode == "overlap" else 2) if persistent else 1)
        assert int((status != 0).sum()) == 0
        outs.append(st[..., :25].cpu().numpy())
    for t in (1, 29, 31, 35, 61, T):  # all solves stop at |update| <= 1e-8: agreement at that level
        assert rel_l2(outs[0][t], outs[2][t]) < 1e-7
        assert rel_l2(outs[1][t], outs[2][t]) < 1e-7


@pytest.mark.parametrize("N,dtype", [(20, "f64"), (20, "f32"), (6, "f64")])
def test_step_batch_matches_simulate(torch_cuda, shooting_mode, N, dtype):
    """kr_step_batch - the loop body of knode.simulate (knode.py:70-100) on its own: T steps taken one call at a time
    (first with prev = cur as knode.py:65-66, then with one and two older states for the time extrapolation of the
    start values) reach the states kr_simulate_batch stores, and both equal the oracle's."""
    torch = torch_cuda
    import cosserat_oracle as orc
    if shooting_mode in ("persistent", "overlap"):
        pytest.skip("kr_step_batch is one launch per call: same kernels as the 'single' / 'multi' parametrisations")
    want = expected_path(shooting_mode, N)
    r = make_robot(None, N)
    h = r._native()
    dt = torch.float64 if dtype == "f64" else torch.float32
    B, T = 5, 9
    ctl = orc.batch_sine_controls(B, T, r.del_t, 99)
    ctl_t = torch.as_tensor(ctl, device="cuda:0").to(dt).contiguous()
    ref = h.new_state(B, dt, n_slots=T + 1)
    h.init_straight(ref[0])
    Gr = torch.zeros((B, 6), dtype=dt, device="cuda:0")
    h.set_option("persistent", 0)
    h.simulate(ctl_t, ref, Gr)
    for use_prev2 in (False, True):
        st = h.new_state(B, dt, n_slots=T + 1)
        h.init_straight(st[0])
        G = torch.zeros((B, 6), dtype=dt, device="cuda:0")
        status = torch.full((B,), -1, dtype=torch.int32, device="cuda:0")
        iters = torch.zeros((B,), dtype=torch.int32, device="cuda:0")
        for t in range(T):
            prev = st[t - 1] if t else st[0]
            prev2 = st[t - 2] if (use_prev2 and t >= 2) else None
            h.step(prev, st[t], st[t + 1], G, ctl_t[:, t].contiguous(), status=status, iters=iters, prev2=prev2)
            assert_path(h, want)
            assert int((status != 0).sum()) == 0 and int(iters.min()) >= 1
        scale = float(ref[T].abs().max())
        tol = (1e-9 if dtype == "f64" else 2e-4) * scale
        assert float((st[T] - ref[T]).abs().max()) < tol
        assert float((G - Gr).abs().max()) < (1e-8 if dtype == "f64" else 1e-3) * max(1.0, float(Gr.abs().max()))
    D = orc.params_for(None, N).derived()
    tip = orc.simulate(D, np.vstack([ctl[2], ctl[2][-1:]]), solver="newton")[1:, :3, -1]
    got = torch.stack([h.tip(st[t + 1])[2] for t in range(T)]).double().cpu().numpy()
    assert rel_l2(got, tip) < (1e-8 if dtype == "f64" else 1e-5)


def test_error_paths(torch_cuda, shooting_mode):
    _once(shooting_mode)
    torch = torch_cuda
    import krod_native as kn
    from knode import setup_robot
    r = make_robot(None, 10)
    with pytest.raises(Exception):
        setup_robot(r, "bogus")
    with pytest.raises(Exception):
        setup_robot(r, None, original=True)
    h = r._native()
    st = h.new_state(2, torch.float64, n_slots=3)
    G = torch.zeros((2, 6), dtype=torch.float64, device="cuda:0")
    tens = torch.ones((2, 4), dtype=torch.float64, device="cuda:0")
    with pytest.raises(kn.KrError):  # aliasing
        h.step(st[0], st[1], st[1], G, tens)
    with pytest.raises(kn.KrError):  # NN requested but never set
        h.step(st[0], st[1], st[2], G, tens, use_nn=True)
    with pytest.raises(kn.KrError):  # host tensor
        h.step(st[0].cpu(), st[1], st[2], G, tens)
    # empty batch is a no-op
    h.step(st[0][:0], st[1][:0], st[2][:0], G[:0], tens[:0])


def test_simulate_legacy_preset(torch_cuda, shooting_mode):
    """prepare.py:35-73 parameters (del_t 0.005, steel rod): forward simulation against the oracle."""
    import cosserat_oracle as orc
    from cosserat_ode import CosseratRod
    from knode import setup_robot_original, simulate
    r = CosseratRod(use_fsolve=True)
    setup_robot_original(r, "damping")
    r.N = 20
    r.compute_intermediate_terms()
    P = orc.RodParams()
    P.N, P.del_t, P.L, P.E, P.r, P.rho = 20, 0.005, 0.4, 209e9, 0.0012, 8000.0
    P.Bbt = np.diag([9e-4] * 3)
    T = 40
    ctl = np.array(orc.calc_controls("sine", 0.5, P.del_t, T))
    want = orc.simulate(P.derived(), ctl, solver="fsolve")
    path = require_path(shooting_mode, 20)
    got = simulate(r, ctl)
    assert_path(r, path)
    assert rel_l2(got[:, :25], want[:, :25]) < 1e-8


def test_keep_predictor_chunked_calls(torch_cuda, shooting_mode):
    """A trajectory advanced by several kr_simulate_batch calls with "keep_predictor" on gives the states of one
    long call (the predictor only chooses where Newton starts), for both launch forms and when they alternate."""
    torch = torch_cuda
    import cosserat_oracle as orc
    r = make_robot(None, 40)
    h = r._native()
    B, T = 8, 48
    dev = "cuda:0"
    ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 21), device=dev).contiguous()

    def run(chunks, keep, flip=False):
        h.set_option("keep_predictor", 0)
        h.set_option("keep_predictor", keep)
        st = h.new_state(B, torch.float64, n_slots=T + 1)
        h.init_straight(st[0])
        G = torch.zeros((B, 6), dtype=torch.float64, device=dev)
        status = torch.full((B, T), -1, dtype=torch.int32, device=dev)
        t0 = 0
        for k, n in enumerate(chunks):
            if flip:
                h.set_option("persistent", k % 2)
            h.simulate(ctl[:, t0:t0 + n].contiguous(), st[t0:], G, status=status[:, t0:t0 + n].contiguous(),
                       prev_init=st[t0 - 1] if t0 else None)
            t0 += n
        torch.cuda.synchronize()
        h.set_option("keep_predictor", 0)
        return st[..., :25].cpu().numpy()

    one = run([T], 0)
    for chunks, keep, flip in (([16, 16, 16], 1, False), ([16, 16, 16], 0, False), ([7, 20, 21], 1, True)):
        got = run(chunks, keep, flip)
        assert rel_l2(got[T], one[T]) < 1e-7 and rel_l2(got[17], one[17]) < 1e-7
    h.set_option("persistent", 1 if shooting_mode in ("persistent", "overlap") else 0)


@pytest.mark.parametrize("seed", range(8))
def test_randomized_parity_vs_oracle(torch_cuda, shooting_mode, seed):
    """Seeded sweep over what the fixed fixtures do not vary together: preset, grid size (odd sizes, sizes around
    the multiple-shooting threshold), batch size, control type and MLP on/off - GPU batch vs the oracle (its tight
    Newton solver; the oracle itself is pinned to the reference's fsolve runs by tests/test_oracle_golden.py)."""
    import cosserat_oracle as orc
    from knode import simulate_batch
    rng = np.random.default_rng(100 + seed)
    mod = [None, "noair", "nsw", "short", "damping", "dampstiff", "lengthstiff", "youngs"][seed % 8]
    N = int(rng.choice([5, 8, 9, 10, 13, 17, 33, 64, 65]))
    B = int(rng.choice([1, 3, 5, 9]))
    T = int(rng.choice([1, 2, 7, 14]))
    kind = ["sine", "step", "random"][seed % 3]
    r = make_robot(mod, N)
    use_mlp = seed % 4 == 3
    mlp = None
    if use_mlp:
        mlp = orc.make_mlp([28, 16, 25], "elu", seed=seed)
        mlp.weights = [w * 0.05 for w in mlp.weights]
        inject(r, mlp)
    ctl = np.empty((B, T, 4))
    for b in range(B):
        if kind == "sine":
            ctl[b] = orc.calc_controls("sine", float(rng.uniform(0.4, 2.5)), r.del_t, T)
        elif kind == "step":
            ctl[b] = np.array(orc.calc_controls("step", float(rng.uniform(0.5, 2.0)), r.del_t, 40))[-T:]
        else:
            ctl[b] = 5.0 + 2.0 * rng.uniform(size=(T, 4))
    path = require_path(shooting_mode, N, mlp)
    out = simulate_batch(r, ctl)
    assert_path(r, path)
    assert np.all(out["status"] == 0), (mod, N, B, T, kind)
    D = orc.params_for(mod, N).derived()
    for b in range(B):
        want = orc.simulate(D, np.vstack([ctl[b], ctl[b][-1:]]), mlp=mlp, solver="newton")
        got = out["traj"][b]
        assert rel_l2(got[:, :, :], want[: T + 1, :25]) < 1e-8, (mod, N, B, T, kind, b)


def test_edge_shapes(torch_cuda, shooting_mode):
    """Empty and extreme shapes: B = 0 and T = 0 are no-ops, the shortest rods the discretisation allows
    (N = 2, 3: one and two segments), and a large batch whose rods equal the same rods solved in small batches,
    through either kernel (the option ms_batch_limit steers the automatic choice)."""
    torch = torch_cuda
    import cosserat_oracle as orc
    from knode import simulate_batch
    dev = "cuda:0"
    r = make_robot(None, 12)
    h = r._native()
    st = h.new_state(4, torch.float64, n_slots=3)
    h.init_straight(st[0])
    G = torch.zeros((4, 6), dtype=torch.float64, device=dev)
    before = st.clone()
    h.simulate(torch.zeros((4, 0, 4), dtype=torch.float64, device=dev), st, G)          # T = 0
    h.simulate(torch.zeros((0, 5, 4), dtype=torch.float64, device=dev), st[:, :0], G[:0])  # B = 0
    torch.cuda.synchronize()
    assert torch.equal(st, before)
    for N in (2, 3):
        rr = make_robot(None, N)
        ctl = orc.batch_sine_controls(3, 5, rr.del_t, 3)
        out = simulate_batch(rr, ctl)
        assert np.all(out["status"] == 0)
        D = orc.params_for(None, N).derived()
        want = orc.simulate(D, np.vstack([ctl[1], ctl[1][-1:]]), solver="newton")
        assert rel_l2(out["traj"][1], want[:6, :25]) < 1e-8
    if shooting_mode == "single":
        return
    # auto mode: multiple shooting unless the batch exceeds "ms_batch_limit" (no limit by default)
    rr = make_robot(None, 20)
    hh = rr._native()
    hh.set_option("ms_mode", -1)
    B, T = 2500, 4
    ctl = orc.batch_sine_controls(B, T, rr.del_t, 11)
    big = simulate_batch(rr, ctl, tip_only=True)
    assert hh.get_option("last_sim_path") in (1, 2) and np.all(big["status"] == 0)
    hh.set_option("ms_batch_limit", 2048)
    big0 = simulate_batch(rr, ctl, tip_only=True)
    assert hh.get_option("last_sim_path") == 0 and np.all(big0["status"] == 0)
    small = simulate_batch(rr, ctl[:7], tip_only=True)
    assert hh.get_option("last_sim_path") in (1, 2)
    assert rel_l2(big["tip"][:7], small["tip"]) < 1e-7 and rel_l2(big0["tip"][:7], small["tip"]) < 1e-7
