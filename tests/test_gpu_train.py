"""GPU parity tests of the KNODE training path (``-m gpu``): the torch-facing
twin ``CosseratRodTorch``, the MFMA MLP forward/backward, the fused loss and
the fused trainer, against golden vectors captured from the reference's own
training arithmetic (tests/golden/train_step.npz) and against plain torch
fp32/fp64 restatements of the same ops on seeded inputs."""
import numpy as np
import pytest

from conftest import load_golden, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


def make_robot(torch, g, H=64, mod="damping", N=10, hist=False):
    from cosserat_ode_torch import CosseratRodTorch
    from knode import setup_robot
    rob = CosseratRodTorch(DEV, H, nn_input_history=hist)
    setup_robot(rob, mod)
    rob.N = N
    rob.compute_intermediate_terms()
    if g is not None:
        with torch.no_grad():
            rob.nn_models[0].weight.copy_(torch.tensor(g["mlp_W0"]))
            rob.nn_models[0].bias.copy_(torch.tensor(g["mlp_b0"]))
            rob.nn_models[2].weight.copy_(torch.tensor(g["mlp_W1"]))
            rob.nn_models[2].bias.copy_(torch.tensor(g["mlp_b1"]))
    return rob


def four_term_loss(torch, pred, target, kp_pred, kp_tgt):
    """physics_train.py:252-259 with this package's quaternion_to_euler."""
    from Utils.transformations import quaternion_to_euler
    mse = torch.nn.MSELoss()
    return (mse(pred[:3][:, kp_pred], target[:3, kp_tgt]) + mse(pred[7:19][:, kp_pred], target[7:19, kp_tgt])
            + mse(quaternion_to_euler(pred[3:7][:, kp_pred]), quaternion_to_euler(target[3:7, kp_tgt]))
            + mse(pred[19:][:, kp_pred], target[19:, kp_tgt - 1]))


def _check_grads_and_post(torch, rob, g, path):
    params = list(rob.nn_models.parameters())
    for i, p in enumerate(params):
        ref = g[f"{path}_grad{i}"]
        got = p.grad.detach().cpu().numpy()
        assert rel_l2(got, ref) < 2e-4, (path, i, rel_l2(got, ref))
    opt = torch.optim.Adam(params, lr=1e-2, weight_decay=0)
    opt.step()
    with torch.no_grad():
        for nm, p in rob.nn_models.named_parameters():
            if "weight" in nm and "layer1" not in nm:
                p.clamp_(min=0)
    for i, p in enumerate(params):
        ref = g[f"{path}_post{i}"]
        got = p.detach().cpu().numpy()
        # Adam's first step is lr*g/(|g|+eps): entries whose gradient is ~0 are ill-conditioned, so
        # compare with an absolute tolerance well below lr
        assert np.mean(np.abs(got - ref) < 2e-4) > 0.995, (path, i)
        assert np.all(got[np.abs(ref) == 0] == 0) or "bias" in path


def test_fast_path_reference_loop(torch_cuda):
    """physics_train.py:313-401 with the reference's own loop shape, our robot underneath."""
    torch = torch_cuda
    g = load_golden("train_step")
    rob = make_robot(torch, g)
    traj = torch.tensor(g["traj"], device=DEV)
    controls = torch.tensor(g["controls"], device=DEV)
    kp = np.array([3, 5, 7, 9])
    ys, zs = traj[:29, :19], traj[:29, 19:]
    yps, zps = torch.cat((ys[:1], ys[:-1])), torch.cat((zs[:1], zs[:-1]))
    grows = rob.parallelGetNextSegmentEuler(traj[1:30], kp, {
        "yh": rob.c1 * ys + rob.c2 * yps, "zh": rob.c1 * zs + rob.c2 * zps, "tendon_tensions": controls[:29]})
    assert grows.shape == (29, 25, 4)
    assert rel_l2(grows.detach().cpu().numpy(), g["fast_pred"]) < 1e-6
    loss = 0
    for t in range(29):
        loss = loss + four_term_loss(torch, grows[t], traj[t + 1], torch.arange(4, device=DEV), torch.tensor(kp, device=DEV))
    loss = loss / 29
    assert abs(loss.item() - float(g["fast_loss"])) < 1e-5 * abs(float(g["fast_loss"]))
    loss.backward()
    _check_grads_and_post(torch, rob, g, "fast")


def test_slow_path_reference_loop(torch_cuda):
    """physics_train.py:215-267 (getNextSegmentEuler for every step, key points [2, 6, 9])."""
    torch = torch_cuda
    g = load_golden("train_step")
    rob = make_robot(torch, g)
    traj = torch.tensor(g["traj"], device=DEV)
    controls = torch.tensor(g["controls"], device=DEV)
    kp = torch.tensor([2, 6, 9], device=DEV)
    loss = 0
    preds = []
    for t in range(29):
        y, z = traj[t, :19], traj[t, 19:]
        yp, zp = (y, z) if t == 0 else (traj[t - 1, :19], traj[t - 1, 19:])
        rob.y, rob.z = y, z
        rob.tendon_tensions = controls[t]
        rob.residualArgs["yh"] = rob.c1 * y + rob.c2 * yp
        rob.residualArgs["zh"] = rob.c1 * z + rob.c2 * zp
        grow = rob.getNextSegmentEuler(traj[t + 1].clone())
        assert grow.shape == (25, 10)
        preds.append(grow.detach().cpu().numpy())
        loss = loss + four_term_loss(torch, grow, traj[t + 1], kp, kp)
    loss = loss / 29
    assert rel_l2(np.array(preds), g["slow_pred"]) < 1e-6
    assert abs(loss.item() - float(g["slow_loss"])) < 1e-5 * abs(float(g["slow_loss"]))
    loss.backward()
    _check_grads_and_post(torch, rob, g, "slow")


@pytest.mark.parametrize("keep_pred", [False, True])
@pytest.mark.parametrize("path,kp", [("fast", [3, 5, 7, 9]), ("slow", [2, 6, 9])])
def test_fused_trainer(torch_cuda, path, kp, keep_pred):
    """KnodeTrainer: fused prediction + loss + backward + Adam + clamp equals the reference epoch - with the loss in
    the epilogue of the forward kernel (kr_mlp_forward_loss, the default) and as a kernel of its own (keep_pred)."""
    torch = torch_cuda
    from krod_train import KnodeTrainer
    g = load_golden("train_step")
    rob = make_robot(torch, g)
    traj = torch.tensor(g["traj"], device=DEV)[None]
    controls = torch.tensor(g["controls"], device=DEV)[None]
    tr = KnodeTrainer(rob, traj, controls, kp, keep_pred=keep_pred)
    loss = tr.loss_and_grads()
    torch.cuda.synchronize()
    assert abs(float(loss.item()) - float(g[f"{path}_loss"])) < 2e-5 * abs(float(g[f"{path}_loss"]))
    pred = tr.predictions().cpu().numpy()
    ref_pred = g[f"{path}_pred"] if path == "fast" else g["slow_pred"][:, :, kp]
    assert rel_l2(pred, ref_pred) < 1e-6
    for i, p in enumerate(rob.nn_models.parameters()):
        assert rel_l2(p.grad.cpu().numpy(), g[f"{path}_grad{i}"]) < 2e-4
    tr.apply_update()  # kr_adam_step: Adam + clamp + gradient zeroing in one launch
    assert float(tr.bucket.flat.abs().max()) == 0.0
    for i, p in enumerate(rob.nn_models.parameters()):
        assert np.mean(np.abs(p.detach().cpu().numpy() - g[f"{path}_post{i}"]) < 2e-4) > 0.995
    # a few more epochs must reduce the loss
    l0 = tr.step()
    for _ in range(20):
        l1 = tr.step()
    assert l1 < l0


@pytest.mark.parametrize("path,kp", [("fast", [3, 5, 7, 9]), ("slow", [2, 6, 9])])
def test_train_epoch_call_vs_reference_epoch(torch_cuda, path, kp):
    """kr_train_epoch itself (ONE tr.step() = mlp_fwd2_kernel + loss epilogue, mlp_bwd2_kernel, train_tail_kernel) from the
    golden weights against the reference's own epoch (physics_train.py:306-408 / :209-304, captured in train_step.npz): the
    loss it logs, the post-Adam + clamp weights, and - through the update, which is lr * sign-like at step 1 - the
    gradients: Adam's first step moves a parameter by lr * g / (|g| + eps), so the moments are checked against the
    reference gradients directly (exp_avg = 0.1 g, exp_avg_sq = 0.001 g^2)."""
    torch = torch_cuda
    from krod_train import KnodeTrainer
    g = load_golden("train_step")
    rob = make_robot(torch, g)
    traj = torch.tensor(g["traj"], device=DEV)[None]
    controls = torch.tensor(g["controls"], device=DEV)[None]
    tr = KnodeTrainer(rob, traj, controls, kp)
    loss = tr.step()
    assert tr.fused_epoch, "the 28 -> 64 -> 25 network must be served by kr_train_epoch"
    assert abs(loss - float(g[f"{path}_loss"])) < 2e-5 * abs(float(g[f"{path}_loss"]))
    off = 0
    for i, p in enumerate(rob.nn_models.parameters()):
        n = p.numel()
        ref_g = g[f"{path}_grad{i}"].reshape(-1)
        m = tr.exp_avg[off:off + n].cpu().numpy()
        v = tr.exp_avg_sq[off:off + n].cpu().numpy()
        assert rel_l2(m, 0.1 * ref_g) < 2e-4, (i, rel_l2(m, 0.1 * ref_g))
        assert rel_l2(v, 0.001 * ref_g ** 2) < 4e-4, (i, rel_l2(v, 0.001 * ref_g ** 2))
        assert np.mean(np.abs(p.detach().cpu().numpy() - g[f"{path}_post{i}"]) < 2e-4) > 0.995
        off += n
    assert float(tr.bucket.flat.abs().max()) == 0.0  # the tail leaves the gradient buffer zeroed


def test_second_trainer_on_the_same_robot_packs_afresh(torch_cuda):
    """a trainer built on a robot whose previous trainer was freed gets the same workspace / parameter
    addresses from torch's caching allocator; its first epoch must still run on ITS weights (the first epoch always packs)."""
    torch = torch_cuda
    from krod_train import KnodeTrainer
    g = load_golden("train_step")
    traj = torch.tensor(g["traj"], device=DEV)[None]
    controls = torch.tensor(g["controls"], device=DEV)[None]
    rob = make_robot(torch, g)
    tr = KnodeTrainer(rob, traj, controls, [3, 5, 7, 9])
    first = tr.step()
    for _ in range(5):
        tr.step()
    ptrs = (tr.ws.data_ptr(), tr.flat_p.data_ptr())
    del tr
    with torch.no_grad():  # back to the golden weights: epoch 1 of the new trainer must reproduce `first`
        rob.nn_models[0].weight.copy_(torch.tensor(g["mlp_W0"]))
        rob.nn_models[0].bias.copy_(torch.tensor(g["mlp_b0"]))
        rob.nn_models[2].weight.copy_(torch.tensor(g["mlp_W1"]))
        rob.nn_models[2].bias.copy_(torch.tensor(g["mlp_b1"]))
    tr2 = KnodeTrainer(rob, traj, controls, [3, 5, 7, 9])
    again = tr2.step()
    assert again == first, (again, first, (tr2.ws.data_ptr(), tr2.flat_p.data_ptr()) == ptrs)


@pytest.mark.parametrize("H", [64, 512])
def test_train_epochs_call_equals_single_epochs(torch_cuda, H):
    """kr_train_epochs (n epochs queued by one call, KnodeTrainer.run) against n x kr_train_epoch (step()): the same losses
    and parameters bit for bit - it is the same launches; also across a mix of run() and step() and a loss log that grows."""
    torch = torch_cuda
    from krod_train import KnodeTrainer
    g = load_golden("train_step")
    traj = torch.tensor(g["traj"], device=DEV)[None].repeat(3, 1, 1, 1)
    controls = torch.tensor(g["controls"], device=DEV)[None].repeat(3, 1, 1)
    robs = [make_robot(torch, g if H == 64 else None, H=H) for _ in range(2)]
    with torch.no_grad():
        for a, b in zip(robs[0].nn_models.parameters(), robs[1].nn_models.parameters()):
            b.copy_(a)
    t1 = KnodeTrainer(robs[0], traj, controls, [3, 5, 7, 9])
    t2 = KnodeTrainer(robs[1], traj, controls, [3, 5, 7, 9])
    t2.loss_log = t2.loss_log[:8].clone()  # (forces the log to grow inside run())
    for _ in range(25):
        t1.step(sync_loss=False)
    t2.run(7)
    t2.step(sync_loss=False)
    t2.run(17)
    torch.cuda.synchronize()
    assert t1.adam_step == t2.adam_step == 25
    assert t1.losses() == t2.losses()
    for a, b in zip(robs[0].nn_models.parameters(), robs[1].nn_models.parameters()):
        assert torch.equal(a, b)
    assert torch.equal(t1.exp_avg, t2.exp_avg) and torch.equal(t1.exp_avg_sq, t2.exp_avg_sq)


def test_native_adam_matches_torch(torch_cuda):
    """kr_adam_step against torch.optim.Adam + clamp over several epochs on the same data, incl. weight decay."""
    torch = torch_cuda
    from krod_train import KnodeTrainer
    g = load_golden("train_step")
    traj = torch.tensor(g["traj"], device=DEV)[None]
    controls = torch.tensor(g["controls"], device=DEV)[None]
    for wd in (0.0, 1e-3):
        robs = [make_robot(torch, g), make_robot(torch, g)]
        trs = [KnodeTrainer(robs[0], traj, controls, [3, 5, 7, 9], weight_decay=wd, native_adam=True),
               KnodeTrainer(robs[1], traj, controls, [3, 5, 7, 9], weight_decay=wd, native_adam=False)]
        for _ in range(12):
            la = trs[0].step()
            lb = trs[1].step()
            assert abs(la - lb) <= 1e-4 * abs(lb)
        for pa, pb in zip(robs[0].nn_models.parameters(), robs[1].nn_models.parameters()):
            assert rel_l2(pa.detach().cpu().numpy(), pb.detach().cpu().numpy()) < 1e-4


def test_device_plateau_schedule_matches_torch(torch_cuda):
    """kr_adam_plateau_step: Adam + clamp + ReduceLROnPlateau with the learning rate kept on the device (no host round
    trip per epoch) against the torch pair physics_train.py:199,206 builds - same loss curve, the same learning rate
    after every epoch (a short patience and a large rate make the schedule bite several times), same weights."""
    torch = torch_cuda
    from krod_train import KnodeTrainer
    g = load_golden("train_step")
    traj = torch.tensor(g["traj"], device=DEV)[None]
    controls = torch.tensor(g["controls"], device=DEV)[None]
    robs = [make_robot(torch, g), make_robot(torch, g)]
    kw = dict(lr=0.2, patience=2, factor=0.5)
    dev_tr = KnodeTrainer(robs[0], traj, controls, [3, 5, 7, 9], **kw)
    ref_tr = KnodeTrainer(robs[1], traj, controls, [3, 5, 7, 9], device_plateau=False, **kw)
    assert dev_tr.device_plateau and not ref_tr.device_plateau
    E = 60
    lrs_ref, losses_ref = [], []
    for e in range(E):
        dev_tr.step(sync_loss=False)  # nothing in here waits for the device
        losses_ref.append(ref_tr.step())
        lrs_ref.append(ref_tr.scheduler.get_last_lr()[0])
    losses_dev = dev_tr.losses()
    assert len(losses_dev) == E
    # the schedule must have fired, and identically: replay torch's scheduler on the DEVICE run's own losses
    probe = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=kw["lr"])
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(probe, "min", patience=kw["patience"], factor=kw["factor"])
    for v in losses_dev:
        sch.step(v)
    assert dev_tr.scheduler.get_last_lr()[0] == pytest.approx(sch.get_last_lr()[0], rel=1e-12)
    assert sch.get_last_lr()[0] < kw["lr"] / 3.9, "the schedule was meant to reduce the rate at least twice"
    sd = dev_tr.scheduler.state_dict()
    assert sd["reductions"] >= 2 and sd["best"] == pytest.approx(min(losses_dev), rel=1e-6)
    # and the two runs stay together while their schedules agree
    same = 0
    while same < E and abs(losses_dev[same] - losses_ref[same]) <= 2e-4 * abs(losses_ref[same]):
        same += 1
    assert same >= 10, (losses_dev[:12], losses_ref[:12])
    assert dev_tr.optimizer_state_dict()["param_groups"][0]["lr"] == pytest.approx(sch.get_last_lr()[0], rel=1e-12)


def test_no_nn_self_consistency(torch_cuda):
    """with the MLP off the predictor reproduces the state the
    simulator produced (same Euler rule), and matches the reference's own output."""
    torch = torch_cuda
    g = load_golden("train_step")
    rob = make_robot(torch, None, mod=None)  # the data was generated with setup_robot(None)
    rob.use_nn = False
    traj = torch.tensor(g["traj"], device=DEV)
    controls = torch.tensor(g["controls"], device=DEV)
    t = 11
    rob.tendon_tensions = controls[t]
    rob.residualArgs["yh"] = rob.c1 * traj[t, :19] + rob.c2 * traj[t - 1, :19]
    rob.residualArgs["zh"] = rob.c1 * traj[t, 19:] + rob.c2 * traj[t - 1, 19:]
    out = rob.getNextSegmentEuler(traj[t + 1].clone()).cpu().numpy()
    nxt = g["traj"][t + 1]
    # fp32 one-step error; the reference's own twin shows 4e-7 max abs on this check
    assert np.max(np.abs(out[:19, 1:] - nxt[:19, 1:])) < 5e-6   # y columns 1..N-1
    assert np.max(np.abs(out[19:, 1:] - nxt[19:, :-1])) < 5e-6  # z shifted by one column
    # and against the reference's run of the same call (other preset: 'damping')
    rob2 = make_robot(torch, None, mod="damping")
    rob2.use_nn = False
    rob2.tendon_tensions = controls[t]
    rob2.residualArgs["yh"] = rob2.c1 * traj[t, :19] + rob2.c2 * traj[t - 1, :19]
    rob2.residualArgs["zh"] = rob2.c1 * traj[t, 19:] + rob2.c2 * traj[t - 1, 19:]
    out2 = rob2.getNextSegmentEuler(traj[t + 1].clone()).cpu().numpy()
    assert rel_l2(out2, g["nonn_pred_t11"]) < 1e-6


@pytest.mark.parametrize("sizes,acts,Q", [
    ([28, 64, 25], ["elu"], 116),
    ([28, 512, 25], ["elu"], 116),
    ([28, 64, 64, 25], ["elu", "elu"], 1000),
    ([53, 64, 25], ["tanh"], 257),
    ([28, 48, 80, 25], ["softplus", "relu"], 3001),
    ([18, 64, 64, 6], ["elu", "elu"], 4099),   # the literal network of BASELINE.json configs[2] (fused kernels)
    ([18, 96, 6], ["tanh"], 300),
    # round 5: mlp_fwd2c_kernel (one hidden chunk per wavefront) with 3 and 5 chunks, several row blocks per workgroup
    ([28, 192, 25], ["elu"], 3001),
    ([28, 320, 25], ["elu"], 20011),
    ([28, 512, 25], ["softplus"], 33333),
])
def test_mlp_forward_backward_vs_torch(torch_cuda, sizes, acts, Q):
    """MFMA GEMM chain against torch (fp64 reference of the same fp32 weights)."""
    torch = torch_cuda
    import torch.nn as nn
    from cosserat_ode_torch import CosseratRodTorch
    amap = {"elu": nn.ELU, "tanh": nn.Tanh, "softplus": nn.Softplus, "relu": nn.ReLU}
    torch.manual_seed(0)
    rob = CosseratRodTorch(DEV, 8, nn_input_history=(sizes[0] == 53))
    mods = []
    for k in range(len(sizes) - 1):
        mods.append(nn.Linear(sizes[k], sizes[k + 1]))
        if k < len(acts):
            mods.append(amap[acts[k]]())
    rob.nn_models = nn.ModuleList(mods).to(DEV)
    x = torch.randn(Q, sizes[0], device=DEV)
    gout = torch.randn(Q, sizes[-1], device=DEV)
    out = rob.forward(x)
    (out * gout).sum().backward()
    got = [p.grad.clone() for p in rob.nn_models.parameters()]
    # fp64 torch reference
    ref_mods = nn.Sequential(*[m for m in rob.nn_models]).double()
    for p in ref_mods.parameters():
        p.grad = None
    ref = ref_mods(x.double())
    (ref * gout.double()).sum().backward()
    assert rel_l2(out.detach().cpu().numpy(), ref.detach().cpu().numpy()) < 3e-6
    for a, p in zip(got, ref_mods.parameters()):
        assert rel_l2(a.cpu().numpy(), p.grad.cpu().numpy()) < 2e-5
    ref_mods.float()


@pytest.mark.parametrize("dims,acts,S,K", [([28, 64, 64, 25], [4, 4, 0], 37, 4), ([28, 512, 25], [4, 0], 41, 3),
                                           ([28, 64, 25], [1, 0], 1, 1), ([28, 40, 40, 25], [2, 2, 0], 300, 4),
                                           ([28, 96, 96, 25], [4, 4, 0], 10, 2),
                                           # round 5: several row blocks per workgroup of mlp_fwd2c_kernel, ragged tail
                                           ([28, 256, 25], [4, 0], 3001, 4), ([28, 512, 25], [4, 0], 9973, 3)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_forward_loss_fused_equals_two_kernels(torch_cuda, dims, acts, S, K, accumulate):
    """kr_mlp_forward_loss (loss in the epilogue of the fused forward kernel; the last shape is one the fused kernels
    do not serve: generic path) against kr_mlp_forward + kr_loss_rows_fwd_bwd on the same rows: loss and d loss / d out,
    row counts that are not multiples of the 32-row blocks, option mlp_grad_accumulate on and off."""
    torch = torch_cuda
    import ctypes as C
    import krod_native as kn
    from cosserat_ode import CosseratRod
    rng = np.random.default_rng(17)
    r = CosseratRod()
    r.N = 10
    r.compute_intermediate_terms()
    h = r._native()
    Q, n = S * K, len(acts)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    Ws = [t(0.3 * rng.standard_normal((dims[k + 1], dims[k])) / np.sqrt(dims[k])) for k in range(n)]
    bs = [t(0.1 * rng.standard_normal(dims[k + 1])) for k in range(n)]
    x = torch.zeros((Q, 32), dtype=torch.float32, device=DEV)
    x[:, :28] = t(rng.standard_normal((Q, 28)))
    base = t(rng.standard_normal((Q, 25)))
    base[:, 3] += 2.0
    tgt = t(rng.standard_normal((Q, 25)))
    tgt[:, 3] += 2.0
    dims_c = (C.c_int32 * (n + 1))(*dims)
    acts_c = (C.c_int32 * n)(*acts)
    Wp = (C.c_void_p * n)(*[w.data_ptr() for w in Ws])
    bp = (C.c_void_p * n)(*[b.data_ptr() for b in bs])
    ws = torch.empty(max(h.lib.kr_mlp_ws_bytes(n, dims_c, Q), 16), dtype=torch.uint8, device=DEV)
    h.set_option("mlp_grad_accumulate", accumulate)
    try:
        out1 = torch.zeros((Q, 32), dtype=torch.float32, device=DEV)
        dout1 = torch.full((Q, 32), 7.0, dtype=torch.float32, device=DEV)
        loss1 = torch.full((1,), 0.5, dtype=torch.float32, device=DEV)
        kn.check(h.lib.kr_mlp_forward(h._h, Q, n, dims_c, acts_c, Wp, bp, kn._ptr(x), 32, kn._ptr(out1), kn._ptr(ws), kn._stream()))
        kn.check(h.lib.kr_loss_rows_fwd_bwd(h._h, S, K, kn._ptr(base), kn._ptr(out1), kn._ptr(tgt), 29.0, None,
                                            kn._ptr(loss1), kn._ptr(dout1), kn._stream()))
        out2 = torch.zeros((Q, 32), dtype=torch.float32, device=DEV)
        dout2 = torch.full((Q, 32), 7.0, dtype=torch.float32, device=DEV)
        loss2 = torch.full((1,), 0.5, dtype=torch.float32, device=DEV)
        kn.check(h.lib.kr_mlp_forward_loss(h._h, S, K, n, dims_c, acts_c, Wp, bp, kn._ptr(x), 32, kn._ptr(base), kn._ptr(tgt),
                                           29.0, kn._ptr(out2), kn._ptr(loss2), kn._ptr(dout2), kn._ptr(ws), kn._stream()))
        torch.cuda.synchronize()
    finally:
        h.set_option("mlp_grad_accumulate", 0)
    l1, l2 = float(loss1), float(loss2)
    assert (l1 > 0.5) == bool(accumulate) or l1 > 0  # accumulate: added to the 0.5 that was there
    assert abs(l1 - l2) < 2e-6 * abs(l1)
    assert float((dout1 - dout2).abs().max()) < 1e-5 * float(dout1.abs().max())  # (fp32: the two kernels contract differently)
    assert float(dout2[:, 25:].abs().max()) == 0.0
    # the backward pass that follows finds the activations the fused forward left in the workspace
    dW1 = [torch.zeros_like(w) for w in Ws]; db1 = [torch.zeros_like(b) for b in bs]
    dWp = (C.c_void_p * n)(*[w.data_ptr() for w in dW1]); dbp = (C.c_void_p * n)(*[b.data_ptr() for b in db1])
    kn.check(h.lib.kr_mlp_backward(h._h, Q, n, dims_c, acts_c, Wp, kn._ptr(x), 32, kn._ptr(dout2), kn._ptr(ws), dWp, dbp, kn._stream()))
    xs = x[:, :28].double().requires_grad_(False)
    a = xs
    Wd = [w.double().requires_grad_(True) for w in Ws]
    bd = [b.double().requires_grad_(True) for b in bs]
    actf = {0: lambda v: v, 1: torch.tanh, 2: torch.nn.functional.softplus, 3: torch.relu, 4: torch.nn.functional.elu}
    for k in range(n):
        a = actf[acts[k]](a @ Wd[k].t() + bd[k])
    (a * dout2[:, :25].double()).sum().backward()
    for k in range(n):
        assert rel_l2(dW1[k].cpu().numpy(), Wd[k].grad.cpu().numpy()) < 2e-5
        assert rel_l2(db1[k].cpu().numpy(), bd[k].grad.cpu().numpy()) < 2e-5


def test_loss_kernel_vs_torch_autograd(torch_cuda):
    """kr_loss_fwd_bwd (incl. the quaternion_to_euler chain rule) against torch autograd."""
    torch = torch_cuda
    import ctypes as C
    import krod_native as kn
    from cosserat_ode import CosseratRod
    rng = np.random.default_rng(5)
    S, K, N = 37, 4, 10
    r = CosseratRod()
    r.N = N
    r.compute_intermediate_terms()
    h = r._native()
    idx = np.array([3, 5, 7, 9], dtype=np.int32)
    base = torch.tensor(rng.standard_normal((S * K, 25)), dtype=torch.float32, device=DEV)
    base[:, 3] += 2.0  # keep quaternions away from zero norm
    out = torch.tensor(0.1 * rng.standard_normal((S * K, 32)), dtype=torch.float32, device=DEV)
    out[:, 25:] = 0
    target = torch.tensor(rng.standard_normal((S, 25, N)), dtype=torch.float32, device=DEV)
    target[:, 3] += 2.0
    idx_t = torch.tensor(idx, device=DEV)
    pred = torch.empty((S * K, 25), dtype=torch.float32, device=DEV)
    dout = torch.empty((S * K, 32), dtype=torch.float32, device=DEV)
    loss = torch.zeros(1, dtype=torch.float32, device=DEV)
    kn.check(h.lib.kr_loss_fwd_bwd(h._h, S, K, kn._ptr(base), kn._ptr(out), kn._ptr(target), kn._ptr(idx_t), 29.0,
                                   kn._ptr(pred), kn._ptr(loss), kn._ptr(dout), kn._stream()))
    o = out.clone().double().requires_grad_(True)
    ds = float(r.ds)
    p = base.double() + torch.cat([ds * o[:, :19], o[:, 19:25]], dim=1)
    p3 = p.reshape(S, K, 25).transpose(1, 2)
    total = 0
    kp = torch.tensor(idx.astype(np.int64), device=DEV)
    for s in range(S):
        total = total + four_term_loss(torch, p3[s].float(), target[s], torch.arange(K, device=DEV), kp)
    total = total / 29
    total.backward()
    assert abs(loss.item() - total.item()) < 2e-5 * abs(total.item())
    assert rel_l2(pred.cpu().numpy(), p.detach().float().cpu().numpy()) < 1e-6
    assert rel_l2(dout[:, :25].cpu().numpy(), o.grad[:, :25].cpu().numpy()) < 5e-5
    assert float(dout[:, 25:].abs().max()) == 0.0
    # pre-gathered target rows (kr_gather_targets + kr_loss_rows_fwd_bwd) give the same numbers bit for bit
    rows = torch.empty((S * K, 25), dtype=torch.float32, device=DEV)
    kn.check(h.lib.kr_gather_targets(h._h, S, K, kn._ptr(target), kn._ptr(idx_t), kn._ptr(rows), kn._stream()))
    want = torch.cat([target[:, :19][:, :, kp], target[:, 19:][:, :, kp - 1]], dim=1).transpose(1, 2).reshape(S * K, 25)
    assert torch.equal(rows, want)
    dout2 = torch.empty_like(dout)
    loss2 = torch.zeros_like(loss)
    for pr in (pred.clone().zero_(), None):
        kn.check(h.lib.kr_loss_rows_fwd_bwd(h._h, S, K, kn._ptr(base), kn._ptr(out), kn._ptr(rows), 29.0,
                                            kn._ptr(pr) if pr is not None else None, kn._ptr(loss2),
                                            kn._ptr(dout2), kn._stream()))
        assert torch.equal(dout2, dout)
        assert abs(loss2.item() - loss.item()) <= 1e-6 * abs(loss.item())  # atomics: order of the block sums
        if pr is not None:
            assert torch.equal(pr, pred)


def test_ode_parallel_and_pickle(torch_cuda):
    torch = torch_cuda
    import io
    gk = load_golden("ode_kat")
    gt = load_golden("ode_torch_kat")
    from cosserat_ode_torch import CosseratRodTorch
    from knode import setup_robot
    rob = CosseratRodTorch(DEV, 64)
    setup_robot(rob, None)
    with torch.no_grad():
        rob.nn_models[0].weight.copy_(torch.tensor(gk["mlp_elu64_W0"]))
        rob.nn_models[0].bias.copy_(torch.tensor(gk["mlp_elu64_b0"]))
        rob.nn_models[2].weight.copy_(torch.tensor(gk["mlp_elu64_W1"]))
        rob.nn_models[2].bias.copy_(torch.tensor(gk["mlp_elu64_b1"]))
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    tf = t(gk["tensions"]) @ rob.tendon_dirs
    dys, z = rob.ODE_parallel(t(gk["y"]), t(gk["yh"]), t(gk["zh"]), tf)
    got = torch.cat([dys, z], 1).detach().cpu().numpy()
    assert rel_l2(got, gt["par_elu64_1"]) < 2e-6
    a, b = rob.ODE(t(gk["y"][3]), t(gk["yh"][3]), t(gk["zh"][3]), tf[3])
    assert rel_l2(torch.cat([a, b]).detach().cpu().numpy(), gt["ser_elu64_1"][3]) < 5e-6
    # torch.save({'robot': robot}) round trip (physics_train.py:165,284)
    buf = io.BytesIO()
    torch.save({"robot": rob}, buf)
    buf.seek(0)
    rob2 = torch.load(buf, weights_only=False)["robot"]
    dys2, z2 = rob2.ODE_parallel(t(gk["y"]), t(gk["yh"]), t(gk["zh"]), tf)
    assert torch.equal(dys2, dys) and torch.equal(z2, z)


def test_reference_checkpoint_runs_on_gpu(torch_cuda):
    """A checkpoint written by the reference classes (tests/golden/ref_checkpoint.pth) loaded through
    krod_checkpoint and evaluated by the HIP kernels reproduces the reference's ODE_parallel output; the same
    file feeds the NumPy-side class exactly as cosserat_ode.py:81-88 / physics_train.py:104-110 do."""
    torch = torch_cuda
    import os
    import krod_checkpoint as kc
    from conftest import GOLDEN
    from cosserat_ode import CosseratRod
    from knode import setup_robot
    g = load_golden("checkpoint")
    path = os.path.join(GOLDEN, "ref_checkpoint.pth")
    rob = kc.load_checkpoint(path, DEV)["robot"]
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    tf = t(g["tens"]) @ rob.tendon_dirs
    dys, z = rob.ODE_parallel(t(g["y"]), t(g["yh"]), t(g["zh"]), tf)
    assert rel_l2(torch.cat([dys, z], 1).detach().cpu().numpy(), g["par"]) < 2e-6
    # NumPy-side robot reading the same file (fp64 evaluation of the fp32-trained weights)
    r = CosseratRod(nn_path=path, use_fsolve=True)
    setup_robot(r, "damping")
    assert len(r.param_ls) == 4 and r.param_ls[0].shape == (32, 28)
    ys, zz = r.ODE(g["y"][0], g["yh"][0], g["zh"][0], (g["tens"][0] @ r.tendon_dirs))
    assert rel_l2(np.concatenate([ys, zz]), g["par"][0]) < 5e-6


def test_evaluate_closed_loop_rollout(torch_cuda):
    """physics_train.py:136-167: live weights of the torch robot are injected into the NumPy-side robot and
    rolled out in closed loop; the rollout is the golden NN-on simulate (sim_nn fixture), the score a DTW."""
    torch = torch_cuda
    import torch.nn as nn
    import krod_eval as ke
    from cosserat_ode import CosseratRod
    from cosserat_ode_torch import CosseratRodTorch
    from knode import setup_robot, simulate
    g = load_golden("sim_nn")
    rob = CosseratRodTorch(DEV, 64)
    with torch.no_grad():  # the weights the reference ran the fixture with (fp64 there; fp32 parameters here)
        rob.nn_models[0].weight.copy_(torch.tensor(g["mlp_elu64_W0"]))
        rob.nn_models[0].bias.copy_(torch.tensor(g["mlp_elu64_b0"]))
        rob.nn_models[2].weight.copy_(torch.tensor(g["mlp_elu64_W1"]))
        rob.nn_models[2].bias.copy_(torch.tensor(g["mlp_elu64_b1"]))
    r_eval = CosseratRod(use_fsolve=True)
    setup_robot(r_eval)
    r_eval.N = int(g["elu64_N"])
    r_eval.compute_intermediate_terms()
    ctl = g["elu64_ctl"]
    r_plain = CosseratRod(use_fsolve=True)
    setup_robot(r_plain)
    r_plain.N = r_eval.N
    r_plain.compute_intermediate_terms()
    ref = simulate(r_plain, ctl)[:, :25]
    dtw, traj = ke.evaluate(r_eval, rob, ctl, ref)
    assert np.isfinite(dtw) and dtw > 0  # the MLP changes the tip path
    assert r_eval.nn_path == "whatever" and len(r_eval.param_ls) == 4
    again = simulate(r_eval, ctl)[: len(traj), :25]
    assert np.array_equal(again, traj)
    # the reference's own NN-on rollout (weights there in fp64: agreement at the fp32 rounding of the parameters)
    assert rel_l2(traj, g["elu64_traj"][: len(traj), :25]) < 1e-5


def test_training_driver_end_to_end(torch_cuda, tmp_path, capsys):
    """train_knode.py (the physics_train.py-shaped driver): data from the true parameters, a model with the wrong
    damping plus the MLP, 60 fused epochs - the loss falls, the lines physics_multitrain.py parses are printed, the
    closed-loop evaluation runs, and the checkpoint is in the reference's layout."""
    torch = torch_cuda
    import re
    import train_knode
    import krod_checkpoint as kc
    path = str(tmp_path / "m.pth")
    loss, dtw = train_knode.main(["sine", "2", "--fast", "--mod", "damping", "--epochs", "60", "--layers", "32",
                                  "--save", path])
    out = capsys.readouterr().out
    assert re.search(r"Epoch (\d+)", out) and re.search(r"Total loss: (.*?), lr (.*?)", out)
    assert "Validation DTW Distance XYZ" in out
    assert len(loss) == 60 and loss[-1] < 0.5 * loss[0]
    assert len(dtw) == 2 and all(np.isfinite(d[0]) for d in dtw)
    ck = kc.load_checkpoint(path, DEV)
    assert len(ck["loss"]) == 60 and len(ck["robot"].nn_models) == 3
    assert ck["robot"].nn_models[0].weight.shape == (32, 28)
    assert float(ck["robot"].nn_models[0].weight.detach().min()) >= 0.0  # the clamp of physics_train.py:299-304


def test_data_parallel_more_ranks_than_trajectories(torch_cuda, tmp_path):
    """train_knode.py under torch.distributed.run with 3 ranks sharing the GPU (gloo) and only 2 trajectories: one rank
    holds an empty shard and must contribute zeros, so the loss curve equals the single-process one (round 1 gave
    idle ranks trajectory 0 again).  Also: the checkpoint carries real Adam state and training resumes from it."""
    import os
    import re
    import subprocess
    import sys
    from conftest import PKG
    torch = torch_cuda
    env = dict(os.environ, PYTHONPATH=PKG, KR_DIST_BACKEND="gloo")
    common = ["sine", "random", "2", "7", "--fast", "--mod", "damping", "--epochs", "31", "--layers", "32", "--no-eval"]
    script = os.path.join(PKG, "train_knode.py")

    def losses(cmd, save):
        out = subprocess.run(cmd + common + ["--save", save], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        vals = [float(m) for m in re.findall(r"Total loss: ([-0-9.e]+)", out.stdout)]
        assert len(vals) == 4, out.stdout
        return vals

    one = losses([sys.executable, script], str(tmp_path / "one.pth"))
    port = str(29600 + os.getpid() % 300)
    three = losses([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                    "--master-addr", "127.0.0.1", "--master-port", port, script], str(tmp_path / "three.pth"))
    assert all(abs(a - b) <= 1e-4 * abs(a) for a, b in zip(one, three)), (one, three)
    # the checkpoint's 'optim' entry is an Adam state_dict a torch.optim.Adam accepts
    import krod_checkpoint as kc
    ck = kc.load_checkpoint(str(tmp_path / "one.pth"), DEV)
    sd = ck["optim"]
    params = list(ck["robot"].nn_models.parameters())
    assert len(sd["state"]) == len(params) == 4
    assert all(float(sd["state"][k]["step"]) == 31 and sd["state"][k]["exp_avg"].shape == params[k].shape
               and float(sd["state"][k]["exp_avg_sq"].abs().sum()) > 0 for k in range(4))
    torch.optim.Adam(params, lr=1e-2).load_state_dict(sd)
    # resuming continues the loss curve: 31 + 10 epochs == 41 epochs in one go (epoch 40's line is the last printed)
    def run(extra, epochs, save):
        cmd = [sys.executable, script] + [c if c != "31" else str(epochs) for c in common] + ["--save", save] + extra
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        return [float(m) for m in re.findall(r"Total loss: ([-0-9.e]+)", out.stdout)]
    full = run([], 41, str(tmp_path / "full.pth"))
    cont = run(["--resume", str(tmp_path / "one.pth")], 10, str(tmp_path / "cont.pth"))
    assert len(full) == 5 and len(cont) == 1  # (epoch numbering restarts after a resume: compare the end states)
    ck_full = kc.load_checkpoint(str(tmp_path / "full.pth"), DEV)
    ck_cont = kc.load_checkpoint(str(tmp_path / "cont.pth"), DEV)
    assert abs(ck_cont["loss"][-1] - ck_full["loss"][-1]) <= 1e-4 * abs(ck_full["loss"][-1])
    for a, b in zip(ck_cont["robot"].nn_models.parameters(), ck_full["robot"].nn_models.parameters()):
        assert rel_l2(a.detach().cpu().numpy(), b.detach().cpu().numpy()) < 1e-4


@pytest.mark.parametrize("N", [10, 12])
def test_estimate_state_on_device(torch_cuda, N):
    """kr_estimate_state (four fp64 kernels) against knode_cosserat_realworld/estimate_state.py:158-242 run by the
    reference (fixture estimate_state.npz: N = 10, the size its literal index 9 is meant for, and N = 12, where the
    backward integration wraps into the tip entry), and against the CPU oracle on a longer random input."""
    import krod_estimate as kest
    import estimate_oracle as eor
    from cosserat_ode import CosseratRod
    from knode import setup_robot
    g = load_golden("estimate_state")
    r = CosseratRod(use_fsolve=True)
    setup_robot(r)
    r.N = N
    r.compute_intermediate_terms()
    est = kest.estimate_state(g[f"N{N}_data"], g[f"N{N}_ctl"], r)
    want = g[f"N{N}_est"]
    assert est.shape == want.shape and est.dtype == np.float64
    for rows, name in ((slice(0, 7), "p,h"), (slice(13, 19), "q,w"), (slice(7, 13), "n,m"), (slice(19, 25), "v,u")):
        assert rel_l2(est[:, rows], want[:, rows]) < 1e-9, name
    assert np.allclose(np.asarray(r.vstar, dtype=np.float64), g[f"N{N}_vstar_after"], rtol=0, atol=1e-12)
    # longer input, both implementations: T = 300 steps of jittered poses
    rng = np.random.default_rng(N)
    reps = int(np.ceil(300 / g[f"N{N}_data"].shape[0]))
    data = np.concatenate([g[f"N{N}_data"]] * reps)[:300] + 1e-4 * rng.standard_normal((300, 7, N))
    ctl = np.concatenate([g[f"N{N}_ctl"]] * reps)[:300]
    r2 = CosseratRod(use_fsolve=True)
    setup_robot(r2, "damping")
    r2.N = N
    r2.compute_intermediate_terms()
    r3 = CosseratRod(use_fsolve=True)
    setup_robot(r3, "damping")
    r3.N = N
    r3.compute_intermediate_terms()
    a = kest.estimate_state(data, ctl, r2)
    b = eor.estimate_state(data, ctl, r3)
    for rows in (slice(0, 7), slice(13, 19), slice(7, 13), slice(19, 25)):
        assert rel_l2(a[:, rows], b[:, rows]) < 1e-9
    # a second call on the same robots: the first one left robot.vstar at the re-estimated root strain WITHOUT
    # recomputing Kse_vstar (estimate_state.py:201), so both calls - and a simulate in between - must see the
    # Kse_vstar of the last compute_intermediate_terms()
    kv = np.array(r2.Kse_vstar, dtype=np.float64).copy()
    a2 = kest.estimate_state(data[:120], ctl[:120], r2)
    b2 = eor.estimate_state(data[:120], ctl[:120], r3)
    assert np.array_equal(np.asarray(r2.Kse_vstar, dtype=np.float64), kv)
    for rows in (slice(0, 7), slice(13, 19), slice(7, 13), slice(19, 25)):
        assert rel_l2(a2[:, rows], b2[:, rows]) < 1e-9
        # same input rows, same Kse_vstar: same estimate (away from the end, where np.gradient's one-sided
        # stencil sits at a different row)
        assert rel_l2(a2[:100, rows], a[:100, rows]) < 1e-9
    with pytest.raises(Exception):
        kest.estimate_state(data[:, :, :-1], ctl, r2)
