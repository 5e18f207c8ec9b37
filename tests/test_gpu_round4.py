"""GPU tests (``-m gpu``) added in round 4 for behaviour of the C ABI rather than a new kernel:

  * one handle driven from two streams (the default two-launch simulate shares per-handle scratch: SimArgs::resume, history
    workspaces, predictor images) - calls are ordered on the device, results equal the one-stream run;
  * a predictor image left by an MLP-off call is not loaded by an MLP-on call of the same batch size (its layout depends
    on the predictor's tap count);
  * ``iters`` of ``kr_step_batch`` counts the plain AND the damped phase on every step kernel (knode_rod.h)."""
import os

import numpy as np
import pytest

from conftest import rel_l2
from gpu_helpers import assert_path, expected_path, inject, make_robot, set_mode_env

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _sim(torch, h, ctl, maxit=0, use_nn=False):
    B, T = ctl.shape[0], ctl.shape[1]
    st = h.new_state(B, ctl.dtype, n_slots=T + 1)
    h.init_straight(st[0])
    G = torch.zeros((B, 6), dtype=ctl.dtype, device=DEV)
    tip = torch.empty((B, T, 3), dtype=ctl.dtype, device=DEV)
    status = torch.full((B, T), -1, dtype=torch.int32, device=DEV)
    h.simulate(ctl, st, G, tip=tip, status=status, maxit=maxit, use_nn=use_nn)
    return st, tip, status


@pytest.mark.parametrize("maxit", [0, 2])
def test_one_handle_two_streams(torch_cuda, monkeypatch, maxit):
    """Two kr_simulate_batch calls of one handle queued back to back on DIFFERENT streams, default kernels (overlapped
    launch + take-over launch, which hands the per-rod resume step through the handle's resume buffer).  With maxit = 2
    rods are handed over at different steps in the two calls; unordered, call B's first kernel would overwrite the resume
    steps call A's take-over kernel has not read yet.  The library orders the streams; both results equal their
    one-stream runs bit for bit."""
    torch = torch_cuda
    import cosserat_oracle as orc
    set_mode_env(monkeypatch, "overlap")
    r = make_robot(None, 100)
    h = r._native()
    B, T = 1024, 24
    ctl_a = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 5), device=DEV).contiguous()
    ctl_b = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 6) * 1.3, device=DEV).contiguous()
    ref_a = _sim(torch, h, ctl_a, maxit)
    ref_b = _sim(torch, h, ctl_b, maxit)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(s1):
            got_a = _sim(torch, h, ctl_a, maxit)
        with torch.cuda.stream(s2):
            got_b = _sim(torch, h, ctl_b, maxit)
        torch.cuda.synchronize()
        for got, ref in ((got_a, ref_a), (got_b, ref_b)):
            assert torch.equal(got[2], ref[2])
            assert torch.equal(got[1], ref[1])
            assert torch.equal(got[0][T], ref[0][T])
    assert h.get_option("last_overlap") == 1


def test_wrong_current_device_is_an_error(torch_cuda):
    """The handle belongs to one device; a call from a thread whose current device is another one is refused instead of
    launching into the wrong context (only checkable with more than one device - otherwise the positive path)."""
    torch = torch_cuda
    import krod_native as kn
    r = make_robot(None, 20)
    h = r._native()
    st = h.new_state(2, torch.float64)
    h.init_straight(st)
    if torch.cuda.device_count() > 1:
        with torch.cuda.device(1):
            with pytest.raises(kn.KrError):
                h.init_straight(st)
    torch.cuda.synchronize()


def test_predictor_image_is_keyed_on_the_mlp(torch_cuda, monkeypatch):
    """keep_predictor: an MLP-off call leaves a 3-tap predictor image; the MLP-on call of the same batch size that follows
    uses 5 taps and another layout - it must rebuild its predictor, i.e. give what a fresh handle gives."""
    torch = torch_cuda
    import cosserat_oracle as orc
    set_mode_env(monkeypatch, "persistent")
    N, B, T = 40, 8, 12
    mlp = orc.make_mlp([28, 64, 64, 25], "elu", seed=7)
    ctl = torch.as_tensor(orc.batch_sine_controls(B, T, 0.05, 9), device=DEV).contiguous()
    r = make_robot(None, N)
    inject(r, mlp)
    h = r._native()
    fresh = _sim(torch, h, ctl, use_nn=True)
    r2 = make_robot(None, N)
    inject(r2, mlp)
    h2 = r2._native()
    h2.set_option("keep_predictor", 1)
    _sim(torch, h2, ctl, use_nn=False)          # leaves an MLP-off image for B rods
    again = _sim(torch, h2, ctl, use_nn=True)
    torch.cuda.synchronize()
    assert torch.equal(again[2], fresh[2]) and int((fresh[2] != 0).sum()) == 0
    assert torch.equal(again[1], fresh[1])


@pytest.mark.parametrize("mode,W", [("single", 1), ("multi", 1), ("multi", 2)])
def test_iters_counts_both_phases(torch_cuda, monkeypatch, mode, W):
    """kr_step_batch with an iteration cap of 2 from the straight rod and a large tension jump: plain Newton hits the cap,
    the damped fallback finishes the step - `iters` reports more than the cap on every step kernel (plain + damped), and
    the same status."""
    torch = torch_cuda
    N = 100
    set_mode_env(monkeypatch, mode, waves_per_rod=W)
    r = make_robot(None, N)
    h = r._native()
    B = 4
    st = h.new_state(B, torch.float64, n_slots=2)
    h.init_straight(st[0])
    G = torch.zeros((B, 6), dtype=torch.float64, device=DEV)
    tens = torch.tensor([[9.0, 5.0, 5.0, 9.0]] * B, dtype=torch.float64, device=DEV)
    status = torch.full((B,), -1, dtype=torch.int32, device=DEV)
    iters = torch.zeros((B,), dtype=torch.int32, device=DEV)
    h.step(st[0], st[0], st[1], G, tens, maxit=2, status=status, iters=iters)
    torch.cuda.synchronize()
    assert_path(h, expected_path(mode, N), W) if W == 1 else None
    if W > 1:
        assert h.get_option("last_waves_per_rod") == W
    assert int(status.max()) <= 1
    assert int(iters.min()) > 2, iters.tolist()   # 2 plain iterations + at least one damped one


@pytest.mark.parametrize("act", ["tanh", "softplus", "relu", "elu"])
@pytest.mark.parametrize("dtype,tol", [("f64", 1e-8), ("f32", 1e-5)])
@pytest.mark.parametrize("mode,W", [("persistent", 1), ("multi", 1), ("persistent", 2)])
def test_three_layer_networks_every_activation(torch_cuda, monkeypatch, act, dtype, tol, mode, W):
    """28 -> 64 -> 48 -> 25 with each activation the reference's get_nn_output knows (cosserat_ode.py:92-106) through the
    one-chunk evaluators of round 4 - fp64 base chain (mlp_jvp_tile3) in fp64 sweeps, fp32 base chain on
    v_mfma_f32_4x4x1_16B (mlp_jvp_tile3f) in fp32 sweeps - against the oracle's tight Newton solve.  A second hidden layer
    narrower than 64 exercises the zero padding of both fragment forms."""
    import cosserat_oracle as orc
    from knode import simulate_batch
    set_mode_env(monkeypatch, mode, waves_per_rod=W)
    N, T = 40, 6
    mlp = orc.make_mlp([28, 64, 48, 25], act, seed=13)
    mlp.weights = [w * 1.5 for w in mlp.weights]   # (a correction that matters; at 3 x the oracle's own Newton solve gives up)
    r = make_robot(None, N)
    inject(r, mlp)
    ctl = np.stack([np.array(orc.calc_controls("sine", a, r.del_t, T)) for a in (0.8, 2.0)])
    out = simulate_batch(r, ctl, dtype=dtype)
    h = r._handle
    assert h.get_option("last_sim_path") == (2 if mode == "persistent" else 1) and h.get_option("last_waves_per_rod") == W
    assert np.all(out["status"] == 0)
    D = orc.params_for(None, N).derived()
    plain = orc.simulate(D, np.vstack([ctl[0], ctl[0][-1:]]), solver="newton")
    for b in range(2):
        want, info = orc.simulate(D, np.vstack([ctl[b], ctl[b][-1:]]), mlp=mlp, solver="newton", return_info=True)
        assert np.all(info["ier"][:T] == 1)
        assert rel_l2(out["traj"][b], want[: T + 1, :25]) < tol, (act, dtype, b)
        if b == 0:
            assert rel_l2(plain[: T + 1, :25], want[: T + 1, :25]) > 1e-4


# ---- kr_train_epoch: the epoch as one call (3-4 launches) -------------------------------------------------------------
def _epoch_trainer(torch, layers, M=24, T=12, N=20, kp=(5, 11, 19), seed=5, act="elu"):
    """A KnodeTrainer on M short trajectories simulated by the library itself (MLP off), network 28 -> layers -> 25."""
    import torch.nn as nn
    import cosserat_oracle as orc
    from cosserat_ode_torch import CosseratRodTorch
    from knode import setup_robot, simulate_batch
    from krod_train import KnodeTrainer
    r = make_robot(None, N)
    ctl = orc.batch_sine_controls(M, T, r.del_t, 77)
    traj = torch.as_tensor(simulate_batch(r, ctl, dtype="f32")["traj"][:, :T], device=DEV).float().contiguous()
    rob = CosseratRodTorch(DEV, layers[0])
    setup_robot(rob, "damping")
    rob.N = N
    rob.compute_intermediate_terms()
    torch.manual_seed(seed)
    A = {"elu": nn.ELU, "tanh": nn.Tanh, "softplus": nn.Softplus, "relu": nn.ReLU}[act]
    sizes = [28] + list(layers) + [25]
    mods = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        mods += [nn.Linear(a, b), A()]
    mods = mods[:-1]
    for m in mods:
        if isinstance(m, nn.Linear):
            rob.non_negative_normal_init(m, 0.01, 0.01)
            nn.init.normal_(m.bias, 0.0, 0.01)
    rob.nn_models = nn.ModuleList(mods).to(DEV)
    return KnodeTrainer(rob, traj, torch.as_tensor(ctl, device=DEV).float().contiguous(), list(kp), keep_pred=False)


@pytest.mark.parametrize("layers,act", [([64, 64], "elu"), ([64, 48], "tanh"), ([512], "elu"), ([40], "softplus")])
def test_train_epoch_equals_the_separate_calls(torch_cuda, layers, act):
    """kr_train_epoch (forward + loss, backward, ONE tail launch: slab sum, loss sum, Adam, clamp, plateau schedule,
    fragment update) against kr_mlp_forward_loss + kr_mlp_backward + kr_adam_plateau_step on the same data: the same
    losses and parameters to summation order (the separate path adds its slabs with float atomics)."""
    torch = torch_cuda
    a = _epoch_trainer(torch, layers, act=act)
    b = _epoch_trainer(torch, layers, act=act)
    b.fused_epoch = False
    for _ in range(12):
        a.step(sync_loss=False)
        b.step(sync_loss=False)
    torch.cuda.synchronize()
    assert a.fused_epoch and a.adam_step == 12 and a.scheduler.steps == 12
    la, lb = np.array(a.losses()), np.array(b.losses())
    assert la.shape == lb.shape == (12,) and la[-1] < la[0]
    assert np.max(np.abs(la - lb) / np.abs(lb)) < 5e-6, (la, lb)
    pa, pb = a.flat_p.cpu().numpy(), b.flat_p.cpu().numpy()
    assert np.max(np.abs(pa - pb)) < 2e-6 * max(1.0, np.abs(pb).max())
    assert float(a.bucket.flat.abs().max()) == 0.0                 # gradients and loss slot left zeroed
    assert a.scheduler.get_last_lr() == b.scheduler.get_last_lr()
    assert np.array_equal(a.exp_avg.cpu().numpy() != 0, b.exp_avg.cpu().numpy() != 0)


@pytest.mark.parametrize("layers", [[64, 64], [512]])
def test_train_epoch_is_reproducible_and_its_fragments_track_the_updates(torch_cuda, layers):
    """The tail kernel writes every updated parameter into the MFMA fragment buffers of the next epoch.  Run (a) relies on
    those copies, run (b) has the fragments packed afresh from the parameter vector before every epoch, run (c) splits
    every epoch into the two halves a data-parallel job puts around its all-reduce (phase 1, phase 2): all three must
    agree BIT FOR BIT (fixed summation order everywhere - no atomics in this path)."""
    torch = torch_cuda
    runs = []
    for mode in ("carried", "repacked", "halves"):
        t = _epoch_trainer(torch, layers)
        for _ in range(8):
            if mode == "repacked":
                t.weights_changed()
            if mode == "halves":
                t._epoch_call(1)
                t._epoch_call(2)
            else:
                t.step(sync_loss=False)
        torch.cuda.synchronize()
        runs.append((t.flat_p.cpu().numpy().copy(), np.array(t.losses()), t.exp_avg_sq.cpu().numpy().copy()))
    for other in runs[1:]:
        for x, y in zip(runs[0], other):
            assert np.array_equal(x, y)


def test_train_epoch_notices_parameters_written_from_outside(torch_cuda):
    """load_state_dict / copy_ on the parameters between epochs: the trainer sees torch's version counters move and has
    the fragments packed afresh; a separate kr_adam_plateau_step on the same vector invalidates them in the library."""
    torch = torch_cuda
    a = _epoch_trainer(torch, [64, 64])
    b = _epoch_trainer(torch, [64, 64])
    b.fused_epoch = False
    for t in (a, b):
        for _ in range(3):
            t.step(sync_loss=False)
    sd = {k: v.clone() * 0.5 for k, v in a.robot.nn_models.state_dict().items()}
    for t in (a, b):
        t.robot.nn_models.load_state_dict(sd)
        for _ in range(3):
            t.step(sync_loss=False)
    # mixed use: one epoch through the separate calls on trainer a, then the single call again
    a.fused_epoch = False
    a.step(sync_loss=False)
    a.fused_epoch = True
    a.step(sync_loss=False)
    b.step(sync_loss=False)
    b.step(sync_loss=False)
    torch.cuda.synchronize()
    la, lb = np.array(a.losses()), np.array(b.losses())
    assert np.max(np.abs(la - lb) / np.abs(lb)) < 5e-6, (la, lb)


def test_train_epoch_argument_errors(torch_cuda):
    torch = torch_cuda
    import krod_native as kn
    t = _epoch_trainer(torch, [64, 64])
    with pytest.raises(kn.KrError, match="phase 2 needs"):
        t._epoch_call(2)                       # no phase 1 before it: nothing to scatter into
    with pytest.raises(kn.KrError, match="phase must be"):
        t._epoch_call(3)
    u = _epoch_trainer(torch, [96, 64])        # first hidden layer wider than the three-layer kernels serve
    u.step(sync_loss=False)
    assert u.fused_epoch is False and u.adam_step == 1


def test_train_epoch_random_shapes(torch_cuda):
    """tools/soak_epoch.py: twelve random (trajectories, window, key points, network, activation) draws - row counts from 6
    to ~10^4 incl. fewer rows than a row block and fewer row blocks than a workgroup has wavefronts, hidden widths 8 .. 512 -
    through kr_train_epoch and through the three separate calls: the same losses and parameters."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("soak_epoch", os.path.join(os.path.dirname(__file__), "..", "tools", "soak_epoch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(12, 11, verbose=False) == 0


@pytest.mark.parametrize("act,scale", [("elu", 1.0), ("tanh", 1.5), ("relu", 2.0)])
def test_base_only_storing_sweeps(torch_cuda, act, scale):
    """MLP-on fp64 persistent kernel: storing sweeps evaluate the network at the unperturbed inputs only and are accepted by
    the residual test or a chord update through the last full sweep's factors (option nn_base_only_store, default on).
    Against the same run at a tolerance of 1e-11 the states are as close with the option on as with it off."""
    torch = torch_cuda
    import cosserat_oracle as orc
    N, B, T = 40, 96, 40
    r = make_robot(None, N)
    mlp = orc.make_mlp([28, 64, 64, 25], act, seed=5)
    mlp.weights = [w * scale for w in mlp.weights]
    inject(r, mlp)
    h = r._native()
    assert h.get_option("nn_base_only_store") == 1
    dt = torch.float64
    ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 99), device=DEV).contiguous()

    def run(on, tol=0.0, maxit=0):
        h.set_option("nn_base_only_store", on)
        st = h.new_state(B, dt, n_slots=T + 1)
        h.init_straight(st[0])
        G = torch.zeros((B, 6), dtype=dt, device=DEV)
        status = torch.zeros((B, T), dtype=torch.int32, device=DEV)
        h.simulate(ctl, st, G, ring=False, status=status, use_nn=True, tol=tol, maxit=maxit)
        torch.cuda.synchronize()
        assert h.get_option("last_sim_path") == 2 and int((status != 0).sum()) == 0
        return st.cpu().numpy()[1:, :, :, :25]

    ref = run(0, tol=1e-11, maxit=30)
    den = np.sqrt((ref ** 2).sum(axis=(2, 3)))
    err_on = (np.sqrt(((run(1) - ref) ** 2).sum(axis=(2, 3))) / den).max()
    err_off = (np.sqrt(((run(0) - ref) ** 2).sum(axis=(2, 3))) / den).max()
    h.set_option("nn_base_only_store", 1)
    assert err_on < 5e-8 and err_on < 2.0 * max(err_off, 5e-9), (err_on, err_off)


# ---- round 5: kr_set_mlp packs on the device -----------------------------------------------------------------------------
def test_set_mlp_from_device_pointers_equals_host_pointers(torch_cuda):
    """kr_set_mlp with src_on_device = 1 (one gather launch from the caller's device weights: no device-to-host copy, no
    allocation when the shape repeats) leaves the same packed network as the host-pointer path: kr_mlp_eval_batch and a
    simulate with the MLP inside every sweep agree bit for bit; new weights of the same shape reuse the plan, another shape
    rebuilds it, n_layers = 0 switches the network off and back on."""
    torch = torch_cuda
    import ctypes as C
    import cosserat_oracle as orc
    import krod_native as kn
    from cosserat_ode import CosseratRod
    from knode import setup_robot

    def handle():
        r = CosseratRod(use_fsolve=True)
        setup_robot(r)
        r.N = 40
        r.compute_intermediate_terms()
        return r, r._native()

    (r1, h1), (r2, h2) = handle(), handle()
    x = torch.randn(300, 28, dtype=torch.float64, device=DEV)
    ctl = torch.as_tensor(orc.batch_sine_controls(16, 6, r1.del_t, 3), device=DEV).contiguous()

    def outputs(h):
        out = torch.empty((300, 25), dtype=torch.float64, device=DEV)
        kn.check(h.lib.kr_mlp_eval_batch(h._h, 300, kn._ptr(x), kn._ptr(out), kn.KR_F64, kn._stream()))
        st = h.new_state(16, torch.float64, n_slots=3)
        h.init_straight(st[0])
        G = torch.zeros((16, 6), dtype=torch.float64, device=DEV)
        tip = torch.empty((16, 6, 3), dtype=torch.float64, device=DEV)
        h.simulate(ctl, st, G, ring=True, tip=tip, use_nn=True)
        torch.cuda.synchronize()
        return out, tip

    for sizes, seed in (([28, 64, 64, 25], 1), ([28, 64, 64, 25], 2), ([28, 96, 25], 3), ([28, 64, 64, 25], 4)):
        mlp = orc.make_mlp(sizes, "elu", seed=seed)
        n = len(mlp.weights)
        h1.set_mlp(mlp.weights, mlp.biases, mlp.acts)                       # host pointers
        Wd = [torch.as_tensor(np.ascontiguousarray(w, dtype=np.float32), device=DEV) for w in mlp.weights]
        bd = [torch.as_tensor(np.ascontiguousarray(b, dtype=np.float32), device=DEV) for b in mlp.biases]
        dims = (C.c_int32 * (n + 1))(*sizes)
        acts = (C.c_int32 * n)(*[int(a) for a in mlp.acts])
        Wp = (C.c_void_p * n)(*[w.data_ptr() for w in Wd])
        bp = (C.c_void_p * n)(*[b.data_ptr() for b in bd])
        if seed == 4:  # off and on again: the plan of the shape survives
            kn.check(h2.lib.kr_set_mlp(h2._h, 0, None, None, None, None, 0, kn._stream()))
        kn.check(h2.lib.kr_set_mlp(h2._h, n, dims, acts, Wp, bp, 1, kn._stream()))    # device pointers
        o1, t1 = outputs(h1)
        o2, t2 = outputs(h2)
        assert torch.equal(o1, o2) and torch.equal(t1, t2), sizes
        ref = np.stack([orc.mlp_eval(mlp, xi) for xi in x.cpu().numpy()[:40]])
        assert rel_l2(o1.cpu().numpy()[:40], ref) < 1e-12


@pytest.mark.parametrize("N", [100, 40])
@pytest.mark.parametrize("seed,scale,act", [(7, 1.0, "elu"), (13, 1.5, "tanh"), (3, 0.3, "softplus"), (21, 2.0, "relu")])
def test_fp32_tip_contract_with_base_only_storing_sweeps(torch_cuda, N, seed, scale, act):
    """with base-only storing sweeps (the default) the fp32 MLP-on runs of tools/bo_check.py's
    networks (ELU, tanh x 1.5, softplus x 0.3, ReLU x 2.0; B = 1024; N = 100 and 40) had worst STATE errors of 1e-5 ... 2e-5;
    the contract is on the TIP trajectory: rel L2 over all steps per rod <= 1e-5, worst rod, against the fp64 run of the same
    kernel at a tolerance of 1e-11 - which is itself pinned to the NumPy oracle's tight Newton solve on two rods here."""
    torch = torch_cuda
    import cosserat_oracle as orc
    B, T = 1024, 60
    r = make_robot(None, N)
    mlp = orc.make_mlp([28, 64, 64, 25], act, seed=seed)
    mlp.weights = [w * scale for w in mlp.weights]
    inject(r, mlp)
    h = r._native()
    assert h.get_option("nn_base_only_store") == 1
    ctl_np = orc.batch_sine_controls(B, T, r.del_t, 1237)

    def tips(dt, tol=0.0, maxit=0):
        ctl = torch.as_tensor(ctl_np, device=DEV).to(dt).contiguous()
        st = h.new_state(B, dt, n_slots=3)
        h.init_straight(st[0])
        G = torch.zeros((B, 6), dtype=dt, device=DEV)
        tip = torch.empty((B, T, 3), dtype=dt, device=DEV)
        status = torch.zeros((B, T), dtype=torch.int32, device=DEV)
        h.simulate(ctl, st, G, ring=True, tip=tip, status=status, use_nn=True, tol=tol, maxit=maxit)
        torch.cuda.synchronize()
        assert h.get_option("last_sim_path") == 2 and int((status != 0).sum()) == 0
        return tip.double().cpu().numpy()

    ref = tips(torch.float64, tol=1e-11, maxit=30)
    got = tips(torch.float32)
    err = np.sqrt(((got - ref) ** 2).sum(axis=(1, 2))) / np.sqrt((ref ** 2).sum(axis=(1, 2)))
    assert err.max() <= 1e-5, (N, act, float(err.max()), int(err.argmax()))
    # the reference run itself against the oracle (tight Newton, fp64) on the worst fp32 rod and on rod 0
    D = orc.params_for(None, N).derived()
    for b in {0, int(err.argmax())}:
        want = orc.simulate(D, np.vstack([ctl_np[b, :12], ctl_np[b, 11:12]]), mlp=mlp, solver="newton")[1:, :3, -1]
        assert rel_l2(ref[b, :12], want) < 1e-8, (b, rel_l2(ref[b, :12], want))
