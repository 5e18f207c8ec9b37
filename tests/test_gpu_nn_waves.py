"""GPU parity tests (``-m gpu``) of kr_simulate_batch with the residual MLP inside the sweeps AND several wavefronts per
rod (kr_msw_impl.hpp instantiated with the MLP on, kr_mswn_impl.hpp; ``last_sim_path`` 2, ``last_waves_per_rod`` 2 / 4):
what batches of at most 512 rods run when the robot carries a network (reference cosserat_ode.py:169-184 inside
knode.py:55-102).  Bar: the reference's own MLP fixtures where their grids are long enough to cut (N >= 15 for two
wavefronts), the CPU oracle at N = 100, and the one-wavefront kernel on rough inputs, hard steps, chunked and ring calls.
Everything goes through the C ABI."""
import numpy as np
import pytest

from conftest import load_golden, rel_l2
from gpu_helpers import inject, make_robot, set_mode_env

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _ran(robot_or_handle, W):
    h = robot_or_handle if hasattr(robot_or_handle, "get_option") else robot_or_handle._handle
    assert h.get_option("last_sim_path") == 2, "the persistent kernel was meant to run"
    assert h.get_option("last_waves_per_rod") == W, (h.get_option("last_waves_per_rod"), W)


@pytest.mark.parametrize("name,fixture", [("elu6464", "sim_nn"), ("elu512n24", "sim_more")])
def test_reference_fixtures_two_wavefronts(torch_cuda, monkeypatch, name, fixture):
    """The reference's own runs with an MLP (knode.simulate under fsolve, tests/golden/make_golden.py) on the grids long
    enough for seven sub-intervals: the 28 -> 64 -> 64 -> 25 network at N = 20 and the default 28 -> 512 -> 25 at N = 24."""
    import cosserat_oracle as orc
    from knode import simulate
    set_mode_env(monkeypatch, "persistent", waves_per_rod=2)
    g = load_golden(fixture)
    mlp = orc.mlp_from_arrays(g, f"mlp_{name}")
    r = make_robot(None, int(g[f"{name}_N"]))
    inject(r, mlp)
    traj = simulate(r, g[f"{name}_ctl"])
    _ran(r, 2)
    assert rel_l2(traj[:, :25], g[f"{name}_traj"]) < 1e-8


@pytest.mark.parametrize("W", [2, 4])
@pytest.mark.parametrize("dtype,tol", [("f64", 1e-8), ("f32", 1e-5)])
def test_n100_vs_oracle(torch_cuda, monkeypatch, W, dtype, tol):
    """cfg3's rod (N = 100, 28 -> 64 -> 64 -> 25 ELU) against the CPU oracle's tight Newton solve, three rods with
    different tension amplitudes."""
    import cosserat_oracle as orc
    from knode import simulate_batch
    set_mode_env(monkeypatch, "persistent", waves_per_rod=W)
    N, T = 100, 5
    mlp = orc.make_mlp([28, 64, 64, 25], "elu", seed=7)
    r = make_robot(None, N)
    inject(r, mlp)
    ctl = np.stack([np.array(orc.calc_controls("sine", a, r.del_t, T)) for a in (0.7, 1.5, 2.2)])
    out = simulate_batch(r, ctl, dtype=dtype)
    _ran(r, W)
    assert np.all(out["status"] == 0)
    D = orc.params_for(None, N).derived()
    for b in range(ctl.shape[0]):
        want = orc.simulate(D, np.vstack([ctl[b], ctl[b][-1:]]), mlp=mlp, solver="newton")
        assert rel_l2(out["traj"][b], want[: T + 1, :25]) < tol, (W, dtype, b)


def _mlp(kind):
    import cosserat_oracle as orc
    if kind == "elu6464":
        return orc.make_mlp([28, 64, 64, 25], "elu", seed=7)
    if kind == "elu512":
        m = orc.make_mlp([28, 512, 25], "elu", seed=3)
        m.weights = [w * 0.3 for w in m.weights]
        return m
    if kind == "elu64_128":
        return orc.make_mlp([28, 64, 128, 25], "elu", seed=5)   # two chunks of the second hidden layer
    return orc.make_mlp([28, 64, 25], kind, seed=11)            # tanh / softplus / relu


def _run(torch, h, ctl, dt, ring=False, chunks=None, maxit=0):
    """One or several kr_simulate_batch calls over ctl [B, T, 4]; ring = three-slot state buffer (one call only)."""
    B, T = ctl.shape[0], ctl.shape[1]
    st = h.new_state(B, dt, n_slots=3 if ring else T + 1)
    h.init_straight(st[0])
    G = torch.zeros((B, 6), dtype=dt, device=DEV)
    tip = torch.empty((B, T, 3), dtype=dt, device=DEV)
    status = torch.full((B, T), -1, dtype=torch.int32, device=DEV)
    if ring:
        assert chunks is None
        h.simulate(ctl, st, G, ring=True, tip=tip, status=status, use_nn=True, maxit=maxit)
    else:
        t0 = 0
        for n in (chunks or [T]):
            tp = torch.empty((B, n, 3), dtype=dt, device=DEV)
            sx = torch.full((B, n), -1, dtype=torch.int32, device=DEV)
            h.simulate(ctl[:, t0:t0 + n].contiguous(), st[t0:], G, tip=tp, status=sx, use_nn=True, maxit=maxit,
                       prev_init=st[t0 - 1] if t0 else None)
            tip[:, t0:t0 + n] = tp
            status[:, t0:t0 + n] = sx
            t0 += n
    torch.cuda.synchronize()
    return tip.cpu().numpy(), status.cpu().numpy(), (None if ring else st[..., :25].cpu().numpy())


@pytest.mark.parametrize("kind", ["elu6464", "elu512", "elu64_128", "tanh", "softplus", "relu"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_matches_one_wavefront(torch_cuda, monkeypatch, kind, dtype):
    """Every network shape / activation the base + JVP evaluator serves, on rough inputs (random tensions, steps): two and
    four wavefronts per rod against the one-wavefront kernel - same BVP, same tolerance, so the states agree to what the
    tolerance leaves; ring calls give the tips of full-history calls."""
    torch = torch_cuda
    import cosserat_oracle as orc
    dt = torch.float64 if dtype == "f64" else torch.float32
    N, B, T = 64, 12, 16
    rng = np.random.default_rng(5)
    ctl_np = np.empty((B, T, 4))
    for b in range(B):
        if b % 3 == 0:
            ctl_np[b] = orc.calc_controls("sine", float(rng.uniform(0.5, 2.5)), 0.05, T)
        elif b % 3 == 1:
            ctl_np[b] = np.array(orc.calc_controls("step", float(rng.uniform(0.5, 2.0)), 0.05, 40))[20 - T // 2: 20 + T - T // 2]
        else:
            ctl_np[b] = 5.0 + 2.0 * rng.uniform(size=(T, 4))
    outs = {}
    for W in (1, 2, 4):
        set_mode_env(monkeypatch, "persistent", waves_per_rod=W)
        r = make_robot(None, N)
        inject(r, _mlp(kind))
        h = r._native()
        ctl = torch.as_tensor(ctl_np, device=DEV).to(dt).contiguous()
        tip, status, x = _run(torch, h, ctl, dt)
        _ran(h, W)
        assert np.all(status == 0), (W, np.argwhere(status != 0)[:4])
        assert np.isfinite(x).all()
        tip_ring, status_ring, _ = _run(torch, h, ctl, dt, ring=True)
        _ran(h, W)
        assert np.array_equal(tip_ring, tip) and np.array_equal(status_ring, status)
        outs[W] = (tip, x)
    for W in (2, 4):
        # (fp64: Newton stops at updates below 1e-8 of an O(1) state, the MLP-on Jacobian is approximate: 1e-6;
        #  fp32: the 1e-5 contract)
        assert np.max(np.abs(outs[W][1] - outs[1][1])) < (1e-6 if dtype == "f64" else 2e-4), (W, kind)


@pytest.mark.parametrize("W", [2, 4])
def test_chunked_calls(torch_cuda, monkeypatch, W):
    """A trajectory advanced by several calls (prev_init = the state before the first of a call, "keep_predictor" on: one
    predictor image per wavefront of a rod travels through HBM) gives the states of one long call."""
    torch = torch_cuda
    import cosserat_oracle as orc
    set_mode_env(monkeypatch, "persistent", waves_per_rod=W)
    N, B, T = 100, 6, 24
    r = make_robot(None, N)
    inject(r, _mlp("elu6464"))
    h = r._native()
    ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 31), device=DEV).contiguous()
    _, s_one, one = _run(torch, h, ctl, torch.float64)
    _ran(h, W)
    for keep in (1, 0):
        h.set_option("keep_predictor", keep)
        _, s_got, got = _run(torch, h, ctl, torch.float64, chunks=[7, 1, 16])
        _ran(h, W)
        assert np.all(s_got == 0) and np.all(s_one == 0)
        assert rel_l2(got[T], one[T]) < 1e-7 and rel_l2(got[8], one[8]) < 1e-7
    h.set_option("keep_predictor", 0)


def test_hard_step_status_with_mlp(torch_cuda, monkeypatch):
    """An iteration cap plain Newton cannot meet on a step input: the several-wavefront kernels fall back to damped
    single shooting on wavefront 0 WITH the network (msw_ss_damped), so the status of a step does not depend on the
    kernel the batch size selects - as with the MLP off (test_gpu_msw.py)."""
    torch = torch_cuda
    import cosserat_oracle as orc
    dt = torch.float64
    N, B, T = 100, 3, 6
    base = np.array(orc.calc_controls("step", 3.0, 0.05, T), dtype=np.float64)
    ctl_np = np.stack([base * s for s in (1.0, 1.3, 0.8)])
    outs = {}
    for W in (1, 2, 4):
        set_mode_env(monkeypatch, "persistent", waves_per_rod=W)
        r = make_robot(None, N)
        inject(r, _mlp("elu6464"))
        h = r._native()
        ctl = torch.as_tensor(ctl_np, device=DEV).contiguous()
        tip, status, x = _run(torch, h, ctl, dt, maxit=2)
        _ran(h, W)
        assert np.isfinite(x).all()
        outs[W] = (status, x)
    s1, x1 = outs[1]
    assert np.all((s1 == 0) | (s1 == 1))
    assert np.any(s1 == 0)
    for W in (2, 4):
        sW, xW = outs[W]
        assert np.array_equal(sW, s1), (W, sW, s1)
        if np.all(s1 == 0):
            assert rel_l2(xW[T], x1[T]) < 1e-6


def test_auto_choice_with_mlp(torch_cuda, monkeypatch):
    """Without the override, N = 100 with a network: four wavefronts per rod up to B = 256, two up to B = 512, the
    one-wavefront persistent kernel beyond (a rod's wavefronts need a SIMD each: kr_mswn_impl.hpp); a network the base +
    JVP evaluator does not serve (second hidden layer wider than 192) takes one launch per step."""
    torch = torch_cuda
    set_mode_env(monkeypatch, "persistent", waves_per_rod=0)

    def run(h, B, N):
        st = h.new_state(B, torch.float64, n_slots=3)
        h.init_straight(st[0])
        G = torch.zeros((B, 6), dtype=torch.float64, device=DEV)
        ctl = torch.zeros((B, 2, 4), dtype=torch.float64, device=DEV)
        ctl[:, :, 0] = 1.0
        status = torch.full((B, 2), -1, dtype=torch.int32, device=DEV)
        h.simulate(ctl, st, G, ring=True, status=status, use_nn=True)
        torch.cuda.synchronize()
        assert int((status != 0).sum()) == 0

    r = make_robot(None, 100)
    inject(r, _mlp("elu6464"))
    h = r._native()
    for B, want in ((3, 4), (256, 4), (257, 2), (512, 2), (513, 1)):
        run(h, B, 100)
        assert h.get_option("last_sim_path") == 2 and h.get_option("last_waves_per_rod") == want, (B, want)
    r = make_robot(None, 20)   # too short for thirteen sub-intervals, long enough for seven
    inject(r, _mlp("elu6464"))
    h = r._native()
    run(h, 8, 20)
    assert h.get_option("last_sim_path") == 2 and h.get_option("last_waves_per_rod") == 2
    import cosserat_oracle as orc
    r = make_robot(None, 100)
    inject(r, orc.make_mlp([28, 64, 256, 25], "elu", seed=2))
    h = r._native()
    run(h, 8, 100)
    assert h.get_option("last_sim_path") != 2 and h.get_option("last_waves_per_rod") == 1
