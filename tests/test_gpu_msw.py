"""GPU parity tests (``-m gpu``) of the several-wavefronts-per-rod form of the multiple-shooting step kernel
(kr_msw_impl.hpp; path 1 with ``last_waves_per_rod`` 2 or 4): the kernel long rods in small batches run
(BASELINE cfg5, N = 400 at B <= 512).  Same bar as the one-wavefront kernel: the reference's fixtures
(knode.py:46-101 run by tests/golden/make_golden.py), the CPU oracle, batch independence, a root of the shooting
residual, fp32 inside the 1e-5 contract.  Everything goes through the C ABI."""
import numpy as np
import pytest

from conftest import load_golden, rel_l2
from gpu_helpers import assert_path, make_robot, set_mode_env

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(params=[2, 4])
def waves(request, monkeypatch):
    set_mode_env(monkeypatch, "multi", waves_per_rod=request.param)
    return request.param


@pytest.fixture(params=[2, 4])
def waves_persistent(request, monkeypatch):
    """The persistent form (all steps of kr_simulate_batch in one launch, leading slots of the states kept in LDS)."""
    set_mode_env(monkeypatch, "persistent", waves_per_rod=request.param)
    return request.param


def _n400_case(P):
    if P == "1_0":
        g = load_golden("sim_n400")
        return g["ctl"], g["tip"], g["last"], g["ier"]
    g = load_golden("sim_more")
    return g[f"n400_P{P}_ctl"], g[f"n400_P{P}_tip"], g[f"n400_P{P}_last"], g[f"n400_P{P}_ier"]


@pytest.mark.parametrize("P", ["0_5", "1_0", "2_0", "3_0"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_n400_vs_reference(torch_cuda, waves, P, dtype):
    """calc_controls('sine', P), N = 400: tip path and last state against the reference's own run."""
    from knode import simulate_batch
    ctl, tip, last, ier = _n400_case(P)
    assert np.all(ier == 1)
    r = make_robot(None, 400)
    T = len(tip) - 1
    out = simulate_batch(r, ctl[None, :T], dtype=dtype)
    assert_path(r, 1, waves)
    assert np.all(out["status"] == 0)
    got = np.concatenate([out["traj"][0, :1, :3, -1], out["tip"][0]])
    assert rel_l2(got, tip) < (1e-8 if dtype == "f64" else 1e-5)
    assert rel_l2(out["traj"][0, T], last) < (1e-7 if dtype == "f64" else 2e-5)


@pytest.mark.parametrize("N", [27, 100, 131])
@pytest.mark.parametrize("mod", [None, "dampstiff"])
def test_other_grids_vs_oracle(torch_cuda, waves, N, mod):
    """Grid sizes where the sub-intervals are ragged (N - 1 not a multiple of P = 7 / 13), a model-mismatch variant
    of the rod (knode.py:6-53 'dampstiff'; full material matrices: test_full_matrices): 10 steps
    against the C oracle."""
    import cosserat_oracle as orc
    import cosserat_oracle_c as oc
    from knode import simulate_batch
    if N - 1 < 2 * (4 + 3 * (waves - 1)):
        pytest.skip("too few grid points for this many sub-intervals")
    r = make_robot(mod, N)
    T = 10
    ctl = np.array(orc.calc_controls("sine", 2.0, r.del_t, T))
    out = simulate_batch(r, ctl[None], dtype="f64")
    assert_path(r, 1, waves)
    assert np.all(out["status"] == 0)
    tip_c, _, bad = oc.simulate(orc.params_for(mod, N), ctl)
    assert bad == 0
    assert rel_l2(out["tip"][0], tip_c) < 1e-8


def test_full_matrices(torch_cuda, waves):
    """Off-diagonal material matrices (the DIAG = false instantiation): a rod whose Bse / Bbt are full."""
    import cosserat_oracle as orc
    from knode import simulate_batch
    r = make_robot(None, 60)
    rng = np.random.default_rng(5)
    S = rng.standard_normal((3, 3)) * 0.05
    r.Bbt = r.Bbt @ (np.eye(3) + S + S.T)
    r.Bse = r.Bse + 1e-3 * (S + S.T)
    r.compute_intermediate_terms()
    T = 8
    ctl = np.array(orc.calc_controls("sine", 1.0, r.del_t, T))
    out = simulate_batch(r, ctl[None], dtype="f64")
    assert_path(r, 1, waves)
    assert np.all(out["status"] == 0)
    P = orc.params_for(None, 60)
    P.Bbt = np.array(r.Bbt)
    P.Bse = np.array(r.Bse)
    ref = orc.simulate(P.derived(), np.vstack([ctl, ctl[-1:]]), solver="newton")[1:, :3, -1]
    assert rel_l2(out["tip"][0], ref) < 1e-8


def test_batch_properties(torch_cuda, waves):
    """The cfg5 batch shape (N = 400; B = 512 for two wavefronts per rod, 256 for four), fp64 and fp32: every step
    converges, the stored state is a root of the shooting residual, a rod's result does not depend on the batch
    around it, fp32 within 1e-5 of fp64 at the tip, two rods against the C oracle, and the same trajectory as the
    one-wavefront kernel to rounding."""
    torch = torch_cuda
    import cosserat_oracle as orc
    import cosserat_oracle_c as oc
    r = make_robot(None, 400)
    h = r._native()
    B, T = (512 if waves == 2 else 256), 5
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
        assert_path(h, 1, waves)
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
        # one wavefront per rod on the same inputs
        h.set_option("waves_per_rod", 1)
        st1 = h.new_state(8, dt, n_slots=T + 1)
        h.init_straight(st1[0])
        G1 = torch.zeros((8, 6), dtype=dt, device=DEV)
        h.simulate(ctl_t[:8].contiguous(), st1, G1)
        assert_path(h, 1, 1)
        h.set_option("waves_per_rod", waves)
        scale = float(st1[T].abs().max())
        assert float((st1[T] - states[T][:8]).abs().max()) < (1e-9 if dt == torch.float64 else 2e-4) * scale
    for b in range(B):
        e = rel_l2(tips[torch.float32][b], tips[torch.float64][b])
        assert e < 1e-5, (b, e)
    for b in (1, B - 2):
        tip_c, _, bad = oc.simulate(orc.params_for(None, 400), ctl[b])
        assert bad == 0 and rel_l2(tips[torch.float64][b], tip_c) < 1e-8


def test_step_batch(torch_cuda, waves):
    """kr_step_batch on a long rod (the branch of the kernel that extrapolates its start values from the states it
    is handed, with zero, one and two older states): same states as kr_simulate_batch, tips equal the C oracle's."""
    torch = torch_cuda
    import cosserat_oracle as orc
    import cosserat_oracle_c as oc
    r = make_robot(None, 400)
    h = r._native()
    dt = torch.float64
    B, T = 3, 6
    ctl = np.stack([np.array(orc.calc_controls("sine", P, r.del_t, T)) for P in (0.5, 1.0, 3.0)])
    ctl_t = torch.as_tensor(ctl, device=DEV).contiguous()
    ref = h.new_state(B, dt, n_slots=T + 1)
    h.init_straight(ref[0])
    h.simulate(ctl_t, ref, torch.zeros((B, 6), dtype=dt, device=DEV))
    assert_path(h, 1, waves)
    for use_prev2 in (False, True):
        st = h.new_state(B, dt, n_slots=T + 1)
        h.init_straight(st[0])
        G = torch.zeros((B, 6), dtype=dt, device=DEV)
        status = torch.full((B,), -1, dtype=torch.int32, device=DEV)
        for t in range(T):
            prev = st[t - 1] if t else st[0]
            prev2 = st[t - 2] if (use_prev2 and t >= 2) else None
            h.step(prev, st[t], st[t + 1], G, ctl_t[:, t].contiguous(), status=status, prev2=prev2)
            assert_path(h, 1, waves)
            assert int((status != 0).sum()) == 0
        # (two Newton runs from different start values: equal to within the stopping tolerance 1e-8, not to rounding)
        assert float((st[T] - ref[T]).abs().max()) < 1e-8 * float(ref[T].abs().max())
    for b in range(B):
        tip_c, _, bad = oc.simulate(orc.params_for(None, 400), ctl[b])
        got = torch.stack([h.tip(st[t + 1])[b] for t in range(T)]).cpu().numpy()
        assert bad == 0 and rel_l2(got, tip_c) < 1e-8


def _assert_persistent(h, waves):
    assert h.get_option("last_sim_path") == 2 and h.get_option("last_waves_per_rod") == waves


@pytest.mark.parametrize("P", ["0_5", "1_0", "3_0"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_n400_persistent_vs_reference(torch_cuda, waves_persistent, P, dtype):
    """N = 400 in ONE launch for all steps: the records of such a rod do not fit the LDS, so the persistent kernel keeps its
    history records in global memory (msw_sim_kernel<..., GH>, kr_mswn_impl.hpp: launch_msw_gh_sim).  Same fixtures and
    bars as the one-launch-per-step form above."""
    from knode import simulate_batch
    ctl, tip, last, ier = _n400_case(P)
    assert np.all(ier == 1)
    r = make_robot(None, 400)
    T = len(tip) - 1
    out = simulate_batch(r, ctl[None, :T], dtype=dtype)
    _assert_persistent(r._native(), waves_persistent)
    assert np.all(out["status"] == 0)
    got = np.concatenate([out["traj"][0, :1, :3, -1], out["tip"][0]])
    assert rel_l2(got, tip) < (1e-8 if dtype == "f64" else 1e-5)
    assert rel_l2(out["traj"][0, T], last) < (1e-7 if dtype == "f64" else 2e-5)


def test_n400_persistent_batch(torch_cuda, monkeypatch):
    """The cfg5 batch (B = 512, N = 400, two wavefronts per rod) through the persistent long-rod form: ring call, every step
    converged, tips equal to those of the one-launch-per-step kernel to what the tolerance leaves, chunked calls with the
    predictor images carried through HBM equal to one call."""
    import torch
    import cosserat_oracle as orc
    B, N, T = 512, 400, 12
    ctl_np = orc.batch_sine_controls(B, T, 0.05, 77)
    tips = {}
    for persistent in (1, 0):
        set_mode_env(monkeypatch, "persistent" if persistent else "multi", waves_per_rod=0)
        h = make_robot(None, N)._native()
        ctl = torch.as_tensor(ctl_np, device=DEV).contiguous()
        st = h.new_state(B, torch.float64, n_slots=3)
        h.init_straight(st[0])
        G = torch.zeros((B, 6), dtype=torch.float64, device=DEV)
        tip = torch.empty((B, T, 3), dtype=torch.float64, device=DEV)
        status = torch.full((B, T), -1, dtype=torch.int32, device=DEV)
        h.simulate(ctl, st, G, ring=True, tip=tip, status=status)
        torch.cuda.synchronize()
        assert h.get_option("last_sim_path") == (2 if persistent else 1) and h.get_option("last_waves_per_rod") == 2
        assert int((status != 0).sum()) == 0
        tips[persistent] = tip.cpu().numpy()
        if persistent:  # the same in two calls (state history: three slots are not enough to chunk a ring, use full slots)
            h.set_option("keep_predictor", 1)
            st2 = h.new_state(B, torch.float64, n_slots=T + 1)
            h.init_straight(st2[0])
            G2 = torch.zeros((B, 6), dtype=torch.float64, device=DEV)
            tip2 = torch.empty((B, T, 3), dtype=torch.float64, device=DEV)
            for t0, n in ((0, 5), (5, 7)):
                tp = torch.empty((B, n, 3), dtype=torch.float64, device=DEV)
                h.simulate(ctl[:, t0:t0 + n].contiguous(), st2[t0:], G2, tip=tp, prev_init=st2[t0 - 1] if t0 else None)
                tip2[:, t0:t0 + n] = tp
            torch.cuda.synchronize()
            assert h.get_option("last_sim_path") == 2
            assert np.max(np.abs(tip2.cpu().numpy() - tips[1])) < 1e-8
            h.set_option("keep_predictor", 0)
    assert np.max(np.abs(tips[1] - tips[0])) < 1e-8


@pytest.mark.parametrize("N,mod", [(27, None), (100, None), (100, "dampstiff"), (131, None), (200, "dampstiff")])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_persistent_vs_oracle(torch_cuda, waves_persistent, N, mod, dtype):
    """The persistent several-wavefront kernel: 12 steps against the C oracle (fp64 <= 1e-8, fp32 inside the 1e-5
    contract), full trajectory stored, status reported per step."""
    import cosserat_oracle as orc
    import cosserat_oracle_c as oc
    from knode import simulate_batch
    W = waves_persistent
    if N - 1 < 2 * (4 + 3 * (W - 1)):
        pytest.skip("too few grid points for this many sub-intervals")
    r = make_robot(mod, N)
    T = 12
    ctl = np.array(orc.calc_controls("sine", 2.0, r.del_t, T))
    out = simulate_batch(r, ctl[None], dtype=dtype)
    _assert_persistent(r._native(), W)
    assert np.all(out["status"] == 0)
    tip_c, _, bad = oc.simulate(orc.params_for(mod, N), ctl)
    assert bad == 0
    assert rel_l2(out["tip"][0], tip_c) < (1e-8 if dtype == "f64" else 1e-5)
    assert rel_l2(out["traj"][0, 1:, :3, -1], tip_c) < (1e-8 if dtype == "f64" else 1e-5)  # the stored states, too


def test_persistent_chunked_and_ring(torch_cuda, waves_persistent):
    """A trajectory advanced by several calls with "keep_predictor" (one predictor image per wavefront) and in a 3-slot
    ring equals the trajectory of one call; a rod's result does not depend on the batch around it; the one-launch-per-
    step kernel gives the same states to rounding."""
    torch = torch_cuda
    import cosserat_oracle as orc
    W = waves_persistent
    r = make_robot(None, 100)
    h = r._native()
    dt = torch.float64
    B, T = 6, 30
    ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 5), device=DEV).contiguous()
    full = h.new_state(B, dt, n_slots=T + 1)
    h.init_straight(full[0])
    G0 = torch.zeros((B, 6), dtype=dt, device=DEV)
    h.simulate(ctl, full, G0)
    _assert_persistent(h, W)
    # chunks of 10 steps in a ring, predictor handed over
    h.set_option("keep_predictor", 0)
    h.set_option("keep_predictor", 1)
    ring = h.new_state(B, dt, n_slots=3)
    h.init_straight(ring[0])
    G = torch.zeros((B, 6), dtype=dt, device=DEV)
    tips = []
    prev = None
    cur_slot = 0
    for c in range(3):
        st = ring[[cur_slot, (cur_slot + 1) % 3, (cur_slot + 2) % 3]].contiguous()
        tip = torch.empty((B, 10, 3), dtype=dt, device=DEV)
        h.simulate(ctl[:, 10 * c:10 * c + 10].contiguous(), st, G, ring=True, tip=tip, prev_init=prev)
        _assert_persistent(h, W)
        tips.append(tip)
        ring = st
        cur_slot = 10 % 3          # after 10 steps the newest state sits in slot 10 % 3 of the rotated ring
        prev = st[(cur_slot + 2) % 3].clone()
    h.set_option("keep_predictor", 0)
    got = torch.cat(tips, dim=1)
    want = torch.stack([h.tip(full[t + 1]) for t in range(T)], dim=1)
    assert float((got - want).abs().max()) < 1e-8 * float(want.abs().max())
    # batch independence
    one = h.new_state(1, dt, n_slots=T + 1)
    h.init_straight(one[0])
    h.simulate(ctl[2:3].contiguous(), one, torch.zeros((1, 6), dtype=dt, device=DEV))
    assert torch.equal(one[T][0], full[T][2])
    # the one-launch-per-step kernel
    h.set_option("persistent", 0)
    ps = h.new_state(B, dt, n_slots=T + 1)
    h.init_straight(ps[0])
    h.simulate(ctl, ps, torch.zeros((B, 6), dtype=dt, device=DEV))
    assert_path(h, 1, W)
    h.set_option("persistent", 1)
    assert float((ps[T] - full[T]).abs().max()) < 1e-9 * float(full[T].abs().max())
    assert float(full[T][..., 25:].abs().max()) == 0.0


def test_iteration_cap_is_reported(torch_cuda, waves):
    """A step that runs into the iteration cap ends (every wavefront of the workgroup leaves the loop together) and is
    handed to the damped single-shooting fallback on wavefront 0 (cap 8 x maxit, like the one-wavefront kernels): the
    step either converges there or reports status 1 with its last iterate streamed out; with enough iterations the
    same inputs converge without it."""
    torch = torch_cuda
    import cosserat_oracle as orc
    r = make_robot(None, 150)
    h = r._native()
    dt = torch.float64
    ctl = torch.as_tensor(np.array(orc.calc_controls("step", 3.0, r.del_t, 4), dtype=np.float64)[None], device=DEV).contiguous()
    for maxit, ok in ((1, False), (30, True)):
        st = h.new_state(1, dt, n_slots=5)
        h.init_straight(st[0])
        status = torch.full((1, 4), -1, dtype=torch.int32, device=DEV)
        h.simulate(ctl, st, torch.zeros((1, 6), dtype=dt, device=DEV), status=status, maxit=maxit)
        torch.cuda.synchronize()
        assert_path(h, 1, waves)
        assert bool(torch.isfinite(st).all())
        if ok:
            assert int((status != 0).sum()) == 0
        else:
            assert int((status > 1).sum()) == 0 and int((status < 0).sum()) == 0


@pytest.mark.parametrize("N,persistent", [(400, 0), (100, 1)])
def test_hard_step_status_does_not_depend_on_the_kernel(torch_cuda, monkeypatch, N, persistent):
    """A step input with an iteration cap plain Newton cannot meet: the several-wavefront kernels (W = 2, 4) fall back to
    damped single shooting exactly like the one-wavefront kernel, so the same rods report the same status whatever
    the batch size selects (the reference's trust-region fsolve, knode.py:89, has no such dependence either), and
    the states agree where the step converged."""
    torch = torch_cuda
    import cosserat_oracle as orc
    dt = torch.float64
    B, T = 3, 6
    base = np.array(orc.calc_controls("step", 3.0, 0.05, T), dtype=np.float64)
    ctl_np = np.stack([base * s for s in (1.0, 1.3, 0.8)])
    outs = {}
    for W in (1, 2, 4):
        set_mode_env(monkeypatch, "persistent" if persistent else "multi", waves_per_rod=W)
        r = make_robot(None, N)
        h = r._native()
        ctl = torch.as_tensor(ctl_np, device=DEV).contiguous()
        st = h.new_state(B, dt, n_slots=T + 1)
        h.init_straight(st[0])
        status = torch.full((B, T), -1, dtype=torch.int32, device=DEV)
        h.simulate(ctl, st, torch.zeros((B, 6), dtype=dt, device=DEV), status=status, maxit=2)
        torch.cuda.synchronize()
        assert h.get_option("last_waves_per_rod") == W
        assert bool(torch.isfinite(st).all())
        outs[W] = (status.cpu().numpy(), st[..., :25].cpu().numpy())
    s1, x1 = outs[1]
    assert np.all((s1 == 0) | (s1 == 1))
    for W in (2, 4):
        sW, xW = outs[W]
        assert np.array_equal(sW, s1), (W, sW, s1)
        if np.all(s1 == 0):
            assert rel_l2(xW[T], x1[T]) < 1e-7


def test_auto_choice(torch_cuda, monkeypatch):
    """Without the override: N = 400 takes four wavefronts per rod up to B = 256, two up to B = 512 (one launch for all
    steps, history records in global memory), one beyond (one launch per step); N = 100 runs the persistent several-wavefront kernel with four up to
    B = 256, two up to B = 512 and the persistent one-wavefront kernel beyond; N = 20 always the latter."""
    torch = torch_cuda
    set_mode_env(monkeypatch, "persistent", waves_per_rod=0)

    def run(h, B):
        st = h.new_state(B, torch.float64, n_slots=3)
        h.init_straight(st[0])
        G = torch.zeros((B, 6), dtype=torch.float64, device=DEV)
        ctl = torch.zeros((B, 2, 4), dtype=torch.float64, device=DEV)
        ctl[:, :, 0] = 1.0
        status = torch.full((B, 2), -1, dtype=torch.int32, device=DEV)
        h.simulate(ctl, st, G, status=status)
        assert int((status != 0).sum()) == 0

    h = make_robot(None, 400)._native()
    for B, want in ((1, 4), (256, 4), (257, 2), (512, 2), (513, 1)):
        run(h, B)
        if want > 1:
            assert h.get_option("last_sim_path") == 2 and h.get_option("last_waves_per_rod") == want, (B, want)
        else:
            assert_path(h, 1, want)
    h = make_robot(None, 100)._native()
    for B, want in ((4, 4), (256, 4), (257, 2), (512, 2), (513, 1)):
        run(h, B)
        assert h.get_option("last_sim_path") == 2 and h.get_option("last_waves_per_rod") == want, (B, want)
    h = make_robot(None, 20)._native()
    run(h, 4)
    assert h.get_option("last_sim_path") == 2 and h.get_option("last_waves_per_rod") == 1


@pytest.mark.parametrize("N,kind,dtype", [(100, "sine", "f64"), (64, "random", "f64"), (40, "jumps", "f64"), (400, "sine", "f64"),
                                          (200, "random", "f64"), (100, "sine", "f32"), (400, "jumps", "f32")])
def test_overlapped_steps_on_several_wavefronts(torch_cuda, waves_persistent, N, kind, dtype):
    """kr_mswo_impl.hpp (option "msw_overlap", default on, fp64): the verifying sweep of step t on spare lanes of the
    Jacobian sweep of step t + 1, on 2 or 4 wavefronts per rod.  Same tips, states and status as the plain persistent
    form on smooth and on rough inputs (fresh random tensions every step: rejected verdicts, chord checks, roll-backs),
    as a trajectory, in a 3-slot ring and advanced by three calls; and the oracle's tips on the smooth case.  Rods whose
    tiles of leading slots do not fit the LDS (N = 200, 400) read them from the states in HBM (the GT instantiations); fp32
    within its 1e-5 contract.  Reference: knode.py:55-102 (what both kernels replace)."""
    torch = torch_cuda
    import cosserat_oracle as orc
    W = waves_persistent
    r = make_robot(None, N)
    h = r._native()
    dt = torch.float64 if dtype == "f64" else torch.float32
    f32 = dtype == "f32"
    B, T = 12, 24
    rng = np.random.default_rng(N)
    if kind == "sine":
        ctl_np = orc.batch_sine_controls(B, T, r.del_t, 5)
    elif kind == "random":
        ctl_np = 5.0 + 5.0 * rng.uniform(size=(B, T, 4))
    else:
        ctl_np = np.full((B, T, 4), 5.0)
        for b in range(B):
            ctl_np[b, rng.integers(1, T):, rng.integers(0, 4)] += rng.uniform(-2, 2)
    ctl = torch.as_tensor(ctl_np, device=DEV).to(dt).contiguous()

    def run(overlap, mode):
        h.set_option("msw_overlap", overlap)
        tip = torch.empty((B, T, 3), dtype=dt, device=DEV)
        status = torch.full((B, T), -1, dtype=torch.int32, device=DEV)
        G = torch.zeros((B, 6), dtype=dt, device=DEV)
        ran = []
        if mode == "chunks":
            h.set_option("keep_predictor", 1)
            st = h.new_state(B, dt, n_slots=T + 1)
            h.init_straight(st[0])
            for c in range(3):
                a, b = 8 * c, 8 * c + 8
                tp = torch.empty((B, 8, 3), dtype=dt, device=DEV)
                ss = torch.full((B, 8), -1, dtype=torch.int32, device=DEV)
                h.simulate(ctl[:, a:b].contiguous(), st[a:b + 1], G, tip=tp, status=ss, prev_init=st[a - 1] if a else None)
                ran.append(h.get_option("last_overlap"))
                tip[:, a:b] = tp
                status[:, a:b] = ss
            h.set_option("keep_predictor", 0)
            last = st[T]
        else:
            st = h.new_state(B, dt, n_slots=3 if mode == "ring" else T + 1)
            h.init_straight(st[0])
            h.simulate(ctl, st, G, ring=mode == "ring", tip=tip, status=status)
            ran.append(h.get_option("last_overlap"))
            last = st[T % 3 if mode == "ring" else T]
        _assert_persistent(h, W)
        return tip, status, last.clone(), ran

    try:
        for mode in ("full", "ring", "chunks"):
            t0, s0, l0, ran0 = run(0, mode)
            t1, s1, l1, ran1 = run(1, mode)
            assert all(x == 0 for x in ran0) and all(x == 1 for x in ran1), (mode, ran0, ran1)
            assert int((s0 != 0).sum()) == 0 and torch.equal(s0, s1), mode
            scale = float(t0.abs().max())
            assert float((t0 - t1).abs().max()) < (1e-5 if f32 else 1e-7) * scale, mode   # (both stop at tol 1e-8 / 1e-5)
            assert float((l0[..., :25] - l1[..., :25]).abs().max()) < (1e-3 if f32 else 1e-6) * float(l0[..., :25].abs().max()), mode
            assert float(l1[..., 25:].abs().max()) == 0.0
        if kind == "sine":
            import cosserat_oracle_c as oc
            t1 = run(1, "full")[0].cpu().numpy()
            for b in (0, B - 1):
                tip_c, _, bad = oc.simulate(orc.params_for(None, N), ctl_np[b])
                assert bad == 0 and rel_l2(t1[b].astype(np.float64), tip_c) < (1e-5 if f32 else 1e-8)
    finally:
        h.set_option("msw_overlap", 1)
