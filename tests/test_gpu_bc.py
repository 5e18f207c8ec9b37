"""GPU parity tests (``-m gpu``) of the reference's FULL parameter surface: a tip wrench (cosserat_ode.py:28-29 F_tip,
M_tip -> :206-207), a tilted base quaternion that is not of unit length, p0 != 0 and a base moving with a constant twist
(:44-47 p0, h0, q0, w0 -> :194) and asymmetric tendon directions with a z component (:37-41 -> :195) - all at once,
against runs of the unmodified reference (fixture ``bc.npz``, tests/golden/make_golden.py: gen_bc).  Every kernel family
re-implements the boundary column and the tip rows of its condensation, so every path is exercised: single shooting,
multiple shooting (one launch per step and persistent), the overlapped persistent kernel, 2 and 4 wavefronts per rod,
``kr_step_batch`` and both residual methods, MLP off and on, fp64 and fp32."""
import numpy as np
import pytest

from conftest import load_golden, rel_l2
from gpu_helpers import MODES, PATH_OF_MODE, assert_path, expected_path, inject, make_robot, require_path, set_mode_env

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BC_KEYS = ("F_tip", "M_tip", "p0", "h0", "q0", "w0", "tendon_dirs")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def bc_robot(g, N, mod=None):
    r = make_robot(mod, N)
    for k in BC_KEYS:
        setattr(r, k, np.array(g[f"par_{k}"], dtype=np.float64))
    return r


def _ran(r, mode, W, want):
    if W == 1:
        assert_path(r, want)
    else:
        h = r._handle
        assert h.get_option("last_sim_path") == PATH_OF_MODE[mode] and h.get_option("last_waves_per_rod") == W, \
            (h.get_option("last_sim_path"), h.get_option("last_waves_per_rod"))


def _modes_waves():
    out = [(m, 1) for m in MODES]
    out += [("multi", 2), ("multi", 4), ("persistent", 2), ("persistent", 4)]
    return out


@pytest.mark.parametrize("mode,W", _modes_waves())
@pytest.mark.parametrize("N", [20, 100])
def test_simulate_vs_reference(torch_cuda, monkeypatch, mode, W, N):
    """knode.simulate (fp64) on every step kernel against the reference's trajectory."""
    from knode import simulate
    g = load_golden("bc")
    if W > 1 and N - 1 < 2 * (4 + 3 * (W - 1)):
        pytest.skip("too few grid points for this many sub-intervals")
    want = require_path(mode, N) if W == 1 else None
    set_mode_env(monkeypatch, mode, waves_per_rod=W)
    r = bc_robot(g, N)
    traj = simulate(r, g[f"sim_N{N}_ctl"])
    _ran(r, mode, W, want)
    assert rel_l2(traj[:, :3, -1], g[f"sim_N{N}_tip"]) < 1e-8
    assert rel_l2(traj[-1], g[f"sim_N{N}_last"]) < 1e-7
    if N == 20:
        assert rel_l2(traj[:, :25], g["sim_N20_traj"]) < 1e-8
        # the boundary column carries the parameters (cosserat_ode.py:194)
        assert np.allclose(traj[7, 0:3, 0], g["par_p0"], rtol=0, atol=1e-15)
        assert np.allclose(traj[7, 3:7, 0], g["par_h0"], rtol=0, atol=1e-15)
        assert np.allclose(traj[7, 13:19, 0], np.concatenate([g["par_q0"], g["par_w0"]]), rtol=0, atol=1e-15)
    else:
        assert rel_l2(traj[::10, :25], g["sim_N100_every10"]) < 1e-8


@pytest.mark.parametrize("mode,W", _modes_waves())
def test_simulate_fp32_tip_contract(torch_cuda, monkeypatch, mode, W):
    """fp32 on every path: tip trajectory within BASELINE.json's 1e-5 of the reference (N = 100)."""
    from knode import simulate_batch
    g = load_golden("bc")
    N = 100
    want = require_path(mode, N) if W == 1 else None
    set_mode_env(monkeypatch, mode, waves_per_rod=W)
    r = bc_robot(g, N)
    ctl = g["sim_N100_ctl"]
    T = len(g["sim_N100_tip"]) - 1
    out = simulate_batch(r, ctl[None, :T], dtype="f32")
    _ran(r, mode, W, want)
    assert np.all(out["status"] == 0)
    assert rel_l2(out["tip"][0], g["sim_N100_tip"][1:]) < 1e-5


@pytest.mark.parametrize("mode,W", _modes_waves())
def test_step_input(torch_cuda, monkeypatch, mode, W):
    """A jump in the tensions (calc_controls 'step') on the same rod, N = 40: the acceptance ladders see a hard step."""
    from knode import simulate
    g = load_golden("bc")
    N = 40
    if W > 1 and N - 1 < 2 * (4 + 3 * (W - 1)):
        pytest.skip("too few grid points for this many sub-intervals")
    want = require_path(mode, N) if W == 1 else None
    set_mode_env(monkeypatch, mode, waves_per_rod=W)
    r = bc_robot(g, N)
    traj = simulate(r, g["step_N40_ctl"])
    _ran(r, mode, W, want)
    assert rel_l2(traj[:, :25], g["step_N40_traj"]) < 1e-8


@pytest.mark.parametrize("mode,W", [(m, 1) for m in MODES if m != "overlap"] + [("persistent", 2)])
@pytest.mark.parametrize("name", ["elu6464", "elu64"])
def test_simulate_with_mlp(torch_cuda, monkeypatch, mode, W, name):
    """The residual MLP on (cosserat_ode.py:169-184) with the same boundary / load parameters."""
    import cosserat_oracle as orc
    from knode import simulate
    g = load_golden("bc")
    mlp = orc.mlp_from_arrays(g, f"mlp_{name}")
    N = int(g[f"nn_{name}_N"])
    if W > 1 and N - 1 < 2 * (4 + 3 * (W - 1)):
        pytest.skip("too few grid points for this many sub-intervals")
    want = require_path(mode, N, mlp) if W == 1 else None
    set_mode_env(monkeypatch, mode, waves_per_rod=W)
    r = bc_robot(g, N)
    inject(r, mlp)
    traj = simulate(r, g[f"nn_{name}_ctl"])
    if W == 1:
        assert_path(r, want)
    else:
        assert r._handle.get_option("last_sim_path") == 2 and r._handle.get_option("last_waves_per_rod") == W
    assert rel_l2(traj[:, :25], g[f"nn_{name}_traj"]) < 1e-8


@pytest.mark.parametrize("N", [20, 100])
@pytest.mark.parametrize("scheme", ["euler", "rk4"])
def test_residual_methods(torch_cuda, N, scheme):
    """getResidualEuler / getResidualRK4 (kr_residual_batch): the 6-vector and the mutated y, z."""
    g = load_golden("bc")
    r = bc_robot(g, N)
    tag = f"res_N{N}"
    y0, z0, yp, zp = g[f"{tag}_y"], g[f"{tag}_z"], g[f"{tag}_yp"], g[f"{tag}_zp"]
    yh = r.c1 * y0 + r.c2 * yp
    zh = r.c1 * z0 + r.c2 * zp
    yh_int = 0.5 * (yh[:, :-1] + yh[:, 1:])
    zh_int = 0.5 * (zh[:, :-1] + zh[:, 1:])
    r.tendon_tensions = g[f"{tag}_tens"]
    fn = r.getResidualEuler if scheme == "euler" else r.getResidualRK4
    checked = 0
    for k, G in enumerate(g[f"{tag}_G"]):
        ref_r, ref_y, ref_z = g[f"{tag}_{scheme}_r"][k], g[f"{tag}_{scheme}_y"][k], g[f"{tag}_{scheme}_z"][k]
        if not np.all(np.isfinite(ref_r)):
            continue
        y, z = y0.copy(), z0.copy()
        res = fn(G, y, z, yh, yh_int, zh, zh_int)
        assert rel_l2(y, ref_y) < 1e-10 and rel_l2(z, ref_z) < 1e-10
        assert np.allclose(res, ref_r, rtol=1e-8, atol=1e-10 * np.abs(ref_y[7:13]).max())
        checked += 1
    assert checked or scheme == "rk4"


@pytest.mark.parametrize("mode", ["single", "multi"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_step_batch(torch_cuda, monkeypatch, mode, dtype):
    """kr_step_batch, one launch per time step, states handed over by the caller: reaches the reference's trajectory."""
    torch = torch_cuda
    g = load_golden("bc")
    N = 20
    set_mode_env(monkeypatch, mode)
    want = expected_path(mode, N)
    r = bc_robot(g, N)
    h = r._native()
    dt = torch.float64 if dtype == "f64" else torch.float32
    ctl = np.asarray(g["sim_N20_ctl"], dtype=np.float64)
    T = 12
    ctl_t = torch.as_tensor(ctl[None, :T], device=DEV).to(dt).contiguous()
    st = h.new_state(1, dt, n_slots=T + 1)
    h.init_straight(st[0])
    G = torch.zeros((1, 6), dtype=dt, device=DEV)
    status = torch.full((1,), -1, dtype=torch.int32, device=DEV)
    iters = torch.zeros((1,), dtype=torch.int32, device=DEV)
    for t in range(T):
        prev = st[t - 1] if t else st[0]
        h.step(prev, st[t], st[t + 1], G, ctl_t[:, t].contiguous(), status=status, iters=iters,
               prev2=st[t - 2] if t >= 2 else None)
        assert_path(h, want)
        assert int(status[0]) == 0
    ref = g["sim_N20_traj"]
    for t in (1, 5, T):
        y, z = h.unpack(st[t])
        got = torch.cat([y, z], 1)[0].double().cpu().numpy()
        got[19:, -1] = ref[t, 19:, -1]   # z[:, N-1] is never written by a sweep: it holds the previous state's
        assert rel_l2(got, ref[t]) < (1e-8 if dtype == "f64" else 3e-5), (t, dtype)
    tip = torch.stack([h.tip(st[t + 1])[0] for t in range(T)]).double().cpu().numpy()
    assert rel_l2(tip, g["sim_N20_tip"][1:T + 1]) < (1e-8 if dtype == "f64" else 1e-5)


def test_batch_mixed_with_default_rods(torch_cuda, monkeypatch):
    """Two robots with different parameter sets on two handles, default kernel choice (overlap at N = 100): the tip
    wrench of one does not leak into the other through cached cold tables / predictor images."""
    from knode import simulate
    g = load_golden("bc")
    g0 = load_golden("sim_n100")
    ra, rb = bc_robot(g, 100), make_robot(None, 100)
    ta = simulate(ra, g["sim_N100_ctl"])
    tb = simulate(rb, g0["ctl"])
    ta2 = simulate(ra, g["sim_N100_ctl"])
    assert rel_l2(ta[:, :3, -1], g["sim_N100_tip"]) < 1e-8
    assert rel_l2(tb[:, :3, -1], g0["tip"]) < 1e-8
    assert np.array_equal(ta, ta2)
    # and one handle whose parameters change between calls
    for k in BC_KEYS:
        setattr(rb, k, np.array(g[f"par_{k}"], dtype=np.float64))
    tb2 = simulate(rb, g["sim_N100_ctl"])
    assert rel_l2(tb2[:, :3, -1], g["sim_N100_tip"]) < 1e-8
