"""GPU tests (``-m gpu``) of the overlapped persistent kernel (kr_mso_impl.hpp): the verifying sweep of step t rides on
four spare lanes of the forward-difference sweep of step t + 1.  The kernel is exercised by every ``overlap``
parametrisation of test_gpu_forward.py / test_gpu_configs.py against the reference's fixtures; here are the paths
those do not reach: the hand-over to the second launch (a rod that gives up), rejected verifying sweeps (history
rolled back from HBM), chunked calls on a ring, fp32, one-step calls, and the bench workload against the plain
persistent kernel.  Everything goes through the C ABI."""
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


def _run(torch, h, ctl, dtype, overlap, ring=False, maxit=0, tol=0.0, chunks=None):
    B, T = ctl.shape[0], ctl.shape[1]
    h.set_option("overlap", overlap)
    n_slots = 3 if ring else T + 1
    st = h.new_state(B, dtype, n_slots=n_slots)
    h.init_straight(st[0])
    G = torch.zeros((B, 6), dtype=dtype, device=DEV)
    tip = torch.empty((B, T, 3), dtype=dtype, device=DEV)
    status = torch.full((B, T), -1, dtype=torch.int32, device=DEV)
    if chunks is None:
        h.simulate(ctl, st, G, ring=ring, tip=tip, status=status, maxit=maxit, tol=tol)
        ran = h.get_option("last_overlap")
    else:
        assert ring is False
        t0, ran = 0, 1
        for n in chunks:
            tp = torch.empty((B, n, 3), dtype=dtype, device=DEV)
            sx = torch.full((B, n), -1, dtype=torch.int32, device=DEV)
            h.simulate(ctl[:, t0:t0 + n].contiguous(), st[t0:], G, tip=tp, status=sx, prev_init=st[t0 - 1] if t0 else None)
            ran = min(ran, h.get_option("last_overlap"))
            tip[:, t0:t0 + n] = tp
            status[:, t0:t0 + n] = sx
            t0 += n
    torch.cuda.synchronize()
    return dict(tip=tip.double().cpu().numpy(), status=status.cpu().numpy(), G=G.double().cpu().numpy(),
                states=st.double().cpu().numpy(), ran=ran)


def _sine(B, T, del_t, seed):
    import cosserat_oracle as orc
    return orc.batch_sine_controls(B, T, del_t, seed)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_bench_workload_vs_plain_persistent(torch_cuda, monkeypatch, dtype):
    """B = 1024, N = 100 (the bench workload, 40 steps): same tips and final states as the plain persistent kernel,
    every step converged, and the result of a rod does not depend on the batch it is in."""
    torch = torch_cuda
    set_mode_env(monkeypatch, "overlap")
    dt = torch.float64 if dtype == "f64" else torch.float32
    r = make_robot(None, 100)
    h = r._native()
    B, T = 1024, 40
    ctl = torch.as_tensor(_sine(B, T, r.del_t, 1235), device=DEV).to(dt).contiguous()
    a = _run(torch, h, ctl, dt, 1, ring=True)
    b = _run(torch, h, ctl, dt, 0, ring=True)
    assert a["ran"] == 1 and b["ran"] == 0
    assert np.all(a["status"] == 0) and np.all(b["status"] == 0)
    tol = 1e-8 if dtype == "f64" else 2e-5
    err = np.linalg.norm((a["tip"] - b["tip"]).reshape(B, -1), axis=1) / np.linalg.norm(b["tip"].reshape(B, -1), axis=1)
    assert err.max() < tol
    assert np.abs(a["states"][T % 3] - b["states"][T % 3]).max() < tol * np.abs(b["states"][T % 3]).max()
    assert np.abs(a["G"] - b["G"]).max() < (1e-7 if dtype == "f64" else 1e-3) * max(1.0, np.abs(b["G"]).max())
    c = _run(torch, h, ctl[:5].contiguous(), dt, 1, ring=True)
    assert np.array_equal(c["tip"], a["tip"][:5])
    # oracle on one rod (fp64: the reference's own accuracy class)
    if dtype == "f64":
        import cosserat_oracle as orc
        D = orc.params_for(None, 100).derived()
        cn = ctl[7].double().cpu().numpy()
        ref = orc.simulate(D, np.vstack([cn[:12], cn[11:12]]), solver="newton")[1:, :3, -1]
        assert rel_l2(a["tip"][7, :12], ref) < 1e-8


@pytest.mark.parametrize("kind", ["step", "random", "sine_fast"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_rough_inputs_full_trajectory(torch_cuda, monkeypatch, kind, dtype):
    """Jumps in the controls and fresh random tensions every step (physics_controls.py:22-30): steps need several
    forward-difference sweeps, verifying sweeps get rejected and rolled back.  Every stored state (trajectory mode)
    equals the plain persistent kernel's."""
    torch = torch_cuda
    set_mode_env(monkeypatch, "overlap")
    dt = torch.float64 if dtype == "f64" else torch.float32
    r = make_robot(None, 100)
    h = r._native()
    B, T = 48, 70
    rng = np.random.default_rng(5)
    i = np.arange(1, T + 1)[None, :, None]
    k = np.arange(4)[None, None, :]
    if kind == "sine_fast":
        per = rng.uniform(0.4, 0.6, size=(B, 1, 1))
        ctl = 6.0 + np.sin(2 * np.pi * i * r.del_t / per + k * np.pi / 2)
    elif kind == "step":
        ctl = np.full((B, T, 4), 5.0)
        jump = rng.uniform(0.5, 2.0, size=(B, 1))
        ctl[:, 25:, 0] += jump
        ctl[:, 25:, 3] += jump
        ctl[:, 50:, 1] += 0.5 * jump
    else:
        ctl = 5.0 + 5.0 * rng.uniform(size=(B, T, 4))
    ctl_t = torch.as_tensor(ctl, device=DEV).to(dt).contiguous()
    a = _run(torch, h, ctl_t, dt, 1)
    b = _run(torch, h, ctl_t, dt, 0)
    assert a["ran"] == 1 and b["ran"] == 0
    assert np.all(a["status"] == 0) and np.all(b["status"] == 0)
    tol = 1e-7 if dtype == "f64" else 5e-4
    for t in (1, 24, 26, 30, 51, T):
        assert rel_l2(a["states"][t][..., :25], b["states"][t][..., :25]) < tol, t
    assert float(np.abs(a["states"][..., 25:]).max()) == 0.0  # padding slots
    # the same on a 3-slot ring, where interior records go out lean (leading slots only) and a rolled-back step or a
    # rod taken over by the second launch has to live with that: same tips, and the three states the ring ends with
    # are complete and equal to the trajectory's
    c = _run(torch, h, ctl_t, dt, 1, ring=True)
    assert c["ran"] == 1 and np.all(c["status"] == 0)
    assert np.array_equal(c["tip"], a["tip"])
    for k in (T, T - 1, T - 2):
        assert np.array_equal(c["states"][k % 3], a["states"][k])


def test_hand_over_to_second_launch(torch_cuda, monkeypatch):
    """With an iteration cap of 2 the first steps from the straight rod cannot converge by plain Newton: the overlapped
    kernel gives the rod up at that step and the plain persistent kernel launched behind it takes over there
    (SimArgs::resume) with its warm-start retry and damped fallback.  Status and states must be what the plain kernel
    alone produces with the same cap."""
    torch = torch_cuda
    set_mode_env(monkeypatch, "overlap")
    r = make_robot(None, 100)
    h = r._native()
    B, T = 16, 30
    ctl = torch.as_tensor(_sine(B, T, r.del_t, 77), device=DEV).contiguous()
    a = _run(torch, h, ctl, torch.float64, 1, maxit=2)
    b = _run(torch, h, ctl, torch.float64, 0, maxit=2)
    assert a["ran"] == 1
    # (whether a step ends converged depends on the damped fallback of the second kernel - the same in both runs)
    assert np.all(a["status"] >= 0) and np.all(a["status"] <= 2)
    assert np.array_equal(a["status"] != 0, b["status"] != 0)
    assert np.all(np.isfinite(a["tip"]))
    # from the first failed step on both runs are the plain kernel's: equal to rounding of the start values
    assert rel_l2(a["tip"], b["tip"]) < 1e-6
    # on a ring (lean interior records) the take-over reads what it needs all the same
    ar = _run(torch, h, ctl, torch.float64, 1, maxit=2, ring=True)
    assert np.array_equal(ar["status"], a["status"]) and rel_l2(ar["tip"], a["tip"]) < 1e-9
    for k in (T, T - 1, T - 2):
        assert rel_l2(ar["states"][k % 3][..., :25], a["states"][k][..., :25]) < 1e-9
    # and with the default cap everything converges again on the same handle
    c = _run(torch, h, ctl, torch.float64, 1)
    assert np.all(c["status"] == 0)


def test_chunked_calls_ring_and_single_steps(torch_cuda, monkeypatch):
    """A trajectory advanced in chunks (prev_init handed over, T = 1 calls included) and on a 3-slot ring gives the
    states of one long call."""
    torch = torch_cuda
    set_mode_env(monkeypatch, "overlap")
    r = make_robot(None, 40)
    h = r._native()
    B, T = 9, 36
    ctl = torch.as_tensor(_sine(B, T, r.del_t, 21), device=DEV).contiguous()
    one = _run(torch, h, ctl, torch.float64, 1)
    assert one["ran"] == 1 and np.all(one["status"] == 0)
    for keep in (0, 1):
        h.set_option("keep_predictor", 0)
        h.set_option("keep_predictor", keep)
        ch = _run(torch, h, ctl, torch.float64, 1, chunks=[1, 1, 10, 1, 23])
        h.set_option("keep_predictor", 0)
        assert ch["ran"] == 1 and np.all(ch["status"] == 0)
        assert rel_l2(ch["states"][T][..., :25], one["states"][T][..., :25]) < 1e-7
        assert rel_l2(ch["tip"], one["tip"]) < 1e-7
    ring = _run(torch, h, ctl, torch.float64, 1, ring=True)
    assert np.array_equal(ring["tip"], one["tip"])
    for k in (T, T - 1, T - 2):
        assert np.array_equal(ring["states"][k % 3], one["states"][k])


@pytest.mark.parametrize("mod", ["damping", "short", "default"])
def test_presets_and_other_grids(torch_cuda, monkeypatch, mod):
    """Other parameter sets and grids (ragged interval lengths: N - 1 not a multiple of 4) against the oracle."""
    torch = torch_cuda
    import cosserat_oracle as orc
    set_mode_env(monkeypatch, "overlap")
    for N in (23, 64, 98):  # (fp64: N <= 101 fits four rods per workgroup)
        r = make_robot(mod, N)
        h = r._native()
        B, T = 5, 10
        ctl = _sine(B, T, r.del_t, 3 + N)
        a = _run(torch, h, torch.as_tensor(ctl, device=DEV).contiguous(), torch.float64, 1)
        assert a["ran"] == 1 and np.all(a["status"] == 0)
        D = orc.params_for(mod, N).derived()
        ref = orc.simulate(D, np.vstack([ctl[2], ctl[2][-1:]]), solver="newton")
        assert rel_l2(a["tip"][2], ref[1:, :3, -1]) < 1e-8
        st = a["states"][T][2]  # [N][28] packed: q w v u p h n m
        want = ref[T]           # rows p h n m q w v u
        assert rel_l2(st[:, 12:25].T, want[:13]) < 1e-8 and rel_l2(st[:, 0:6].T, want[13:19]) < 1e-8
        assert rel_l2(st[:, 6:12].T, want[19:25]) < 1e-8


def test_full_matrices_fall_back(torch_cuda, monkeypatch):
    """Non-diagonal damping matrices are not served by the overlapped kernel: the plain persistent kernel runs."""
    torch = torch_cuda
    set_mode_env(monkeypatch, "overlap")
    r = make_robot(None, 40)
    Bbt = np.array(r.Bbt, dtype=np.float64)
    Bbt[0, 1] = Bbt[1, 0] = 0.1 * Bbt[0, 0]
    r.Bbt = Bbt
    r.compute_intermediate_terms()
    h = r._native()
    ctl = torch.as_tensor(_sine(3, 6, r.del_t, 4), device=DEV).contiguous()
    a = _run(torch, h, ctl, torch.float64, 1)
    assert a["ran"] == 0 and h.get_option("last_sim_path") == 2 and np.all(a["status"] == 0)


@pytest.mark.parametrize("mode", ["persistent", "overlap", "multi"])
def test_residual_test_option(torch_cuda, monkeypatch, mode):
    """Option "residual_test" = 0: no storing sweep is accepted from its residual alone - every accepted state carries
    a measured update (Newton or chord) below the tolerance.  Same states as with the default (the option only changes
    how acceptance is decided), for the plain persistent kernel, the overlapped one and one launch per step."""
    torch = torch_cuda
    set_mode_env(monkeypatch, mode)
    r = make_robot(None, 100)
    h = r._native()
    assert h.get_option("residual_test") == 1
    B, T = 32, 45
    ctl = torch.as_tensor(_sine(B, T, r.del_t, 1235), device=DEV).contiguous()
    ov = 1 if mode == "overlap" else 0
    a = _run(torch, h, ctl, torch.float64, ov)
    h.set_option("residual_test", 0)
    b = _run(torch, h, ctl, torch.float64, ov)
    h.set_option("residual_test", 1)
    assert a["ran"] == ov and b["ran"] == ov
    assert np.all(a["status"] == 0) and np.all(b["status"] == 0)
    assert rel_l2(a["states"][T][..., :25], b["states"][T][..., :25]) < 1e-8
    assert rel_l2(a["tip"], b["tip"]) < 1e-9
