"""GPU parity tests (``-m gpu``) added in round 3:

  * BASELINE cfg2 at its size (B = 256, N = 100, T = 200) with the kernel the library picks by itself, against the
    reference's own rods of that draw (fixture round3.npz),
  * BASELINE cfg4 at the size of one of its eight shards (512 trajectories x 29 window steps x 4 key points =
    59 392 rows, the reference's default network 28 -> 512 -> 25, N = 10 and N = 100),
  * ``getResidualRK4`` with midpoint histories that are not the interpolation of yh, zh (cosserat_ode.py:233-235),
  * the side effect of ``knode.simulate`` on ``robot.tendon_tensions`` (knode.py:71),
  * the multi-rank entry points (``bench.py --gpus 2`` and ``train_knode.py`` under torch.distributed.run, gloo
    rendezvous, ranks sharing the one GPU of the box).

Everything goes through the C ABI."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT, load_golden, rel_l2
from gpu_helpers import make_robot, set_mode_env

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


# ---------------------------------------------------------------------------
# cfg2
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_cfg2_full_size_auto_kernel(torch_cuda, monkeypatch, dtype):
    """B = 256 rods, N = 100, T = 200, tensions of SURVEY 8d cfg2 (default_rng(1234)), kernel choice left to the
    library: the persistent kernel with four wavefronts per rod must run, every step converges, and the eight rods the
    reference itself solved with ier == 1 (knode.py:55-102, 50 steps) agree to 1e-8 (fp64) / 1e-5 (fp32)."""
    import cosserat_oracle as orc
    from knode import simulate_batch
    for v in ("KR_MS_MODE", "KR_PERSISTENT", "KR_WAVES_PER_ROD"):
        monkeypatch.delenv(v, raising=False)
    g = load_golden("round3")
    B, T, N = int(g["cfg2_B"]), 200, 100
    r = make_robot(None, N)
    ctl = orc.batch_sine_controls(B, T, r.del_t, int(g["cfg2_seed"]))
    out = simulate_batch(r, ctl, dtype=dtype, tip_only=True)
    h = r._handle
    assert h.get_option("last_sim_path") == 2, "the persistent form must serve cfg2"
    assert h.get_option("last_waves_per_rod") == 4, "B = 256, N = 100: four wavefronts per rod"
    assert np.all(out["status"] == 0)
    assert np.all(np.isfinite(out["tip"]))
    Tf = int(g["cfg2_T"])
    tol = 1e-8 if dtype == "f64" else 1e-5
    for k, b in enumerate(g["cfg2_rods"]):
        # reference entry t (t >= 1) is the state after solve t; simulate_batch's tip[t - 1] is the same solve
        ref = g["cfg2_tip"][k][1:Tf]
        assert rel_l2(out["tip"][b, : Tf - 1], ref) < tol, (int(b), dtype)
    # the long run stays on the attractor of the same motion: bounded, and rods differ
    assert np.abs(out["tip"]).max() < 1.0 and np.std(out["tip"][:, -1, 0]) > 1e-3


# ---------------------------------------------------------------------------
# cfg4 shard
# ---------------------------------------------------------------------------
def _cfg4_controls(M, T, del_t, seed=1236):
    """calc_controls sine / random mix with arguments from default_rng(1236)."""
    from physics_controls import calc_controls
    rng = np.random.default_rng(seed)
    out = []
    for m in range(M):
        if m % 2 == 0:
            out.append(calc_controls("sine", float(rng.uniform(0.5, 3.0)), del_t, T))
        else:
            out.append(calc_controls("random", float(rng.integers(1, 10000)), del_t, T))
    return np.asarray(out, dtype=np.float64)


@pytest.mark.parametrize("N", [10, 100])
def test_cfg4_shard_training_step(torch_cuda, N):
    """One rank's share of BASELINE cfg4: 512 trajectories x train_len 30 (29 window steps) x K = 4 key points =
    59 392 rows through the reference's default network 28 -> 512 -> 25 (cosserat_ode_torch.py:60-62).  Loss and
    every parameter gradient against an fp64 torch restatement of physics_train.py:313-401 on the same rows, the
    update against torch.optim.Adam + the clamp of :299-304."""
    torch = torch_cuda
    import torch.nn as nn
    from cosserat_ode_torch import CosseratRodTorch
    from knode import setup_robot, simulate_batch
    from krod_train import KnodeTrainer
    from Utils.transformations import quaternion_to_euler
    M, T = 512, 30
    kp = [3, 5, 7, 9] if N == 10 else [int(round((N - 1) * k / 9)) for k in (3, 5, 7, 9)]
    rr = make_robot(None, N)
    ctl = _cfg4_controls(M, T, rr.del_t)
    o = simulate_batch(rr, ctl, dtype="f32")
    assert np.all(o["status"] == 0)
    traj = torch.as_tensor(o["traj"][:, :T], device=DEV).float().contiguous()
    controls = torch.as_tensor(ctl, device=DEV).float().contiguous()
    torch.manual_seed(11)
    rob = CosseratRodTorch(DEV, 512)  # physics_train.py:182 with the default --layers
    setup_robot(rob, "damping")
    rob.N = N
    rob.compute_intermediate_terms()
    tr = KnodeTrainer(rob, traj, controls, kp)
    assert tr.Q == 59392 and tr.K == 4 and tr.steps == 29
    # rows the HIP physics kernel produced (kr_next_segment_physics, kr_gather_targets) against the oracle, with
    # calc_controls inputs: x = [y, z, tf] at column key-1 of the teacher-forced next state, base = y + ds * physics
    # (z: physics), target = the true y at column key and z at key-1 (physics_train.py:345-352).  Sine and random trajectories,
    # first / interior / last window step, every key point.
    import cosserat_oracle as orc
    D = orc.setup_params("damping", N).derived()
    xs, bases, tgts = tr.x.cpu().numpy(), tr.base.cpu().numpy(), tr.target_rows[: tr.Q].cpu().numpy()
    tj = traj.cpu().numpy().astype(np.float64)
    for q in (0, 1, 2, 3, 4 * 29 + 5, 30001, 30002, 44444, 59391):
        s_, k_ = divmod(q, 4)
        m_, t_ = divmod(s_, T - 1)
        col = kp[k_] - 1
        Gn = tj[m_, t_ + 1]
        y_, z_ = tj[m_, t_, :19], tj[m_, t_, 19:]
        yp_, zp_ = (y_, z_) if t_ == 0 else (tj[m_, t_ - 1, :19], tj[m_, t_ - 1, 19:])
        yh_, zh_ = D.c1 * y_ + D.c2 * yp_, D.c1 * z_ + D.c2 * zp_
        tf_ = orc.tendon_force(D, ctl[m_, t_].astype(np.float32).astype(np.float64))
        ys_, zz_ = orc.ode(D, Gn[:19, col], yh_[:, col], zh_[:, col], tf_)
        want_x = np.concatenate([Gn[:19, col], zz_, tf_])
        want_b = np.concatenate([Gn[:19, col] + D.ds * ys_, zz_])
        assert np.allclose(xs[q, :28], want_x, rtol=2e-4, atol=2e-5 * np.abs(want_x).max()), q
        assert np.allclose(bases[q], want_b, rtol=2e-4, atol=2e-5 * np.abs(want_b).max()), q
        want_t = np.concatenate([Gn[:19, col + 1], Gn[19:25, col]])   # y rows at the key column, z rows one before (:352)
        assert np.allclose(tgts[q], want_t, rtol=1e-6, atol=1e-7 * np.abs(want_t).max()), q
    w0 = [p.detach().clone() for p in rob.nn_models.parameters()]
    loss = tr.loss_and_grads()
    torch.cuda.synchronize()
    got_loss = float(loss.item())
    got_grads = [p.grad.detach().clone() for p in rob.nn_models.parameters()]
    ref_net = nn.Sequential(nn.Linear(28, 512), nn.ELU(), nn.Linear(512, 25)).to(DEV).double()
    with torch.no_grad():
        for a, b in zip(ref_net.parameters(), w0):
            a.copy_(b.double())
    out = ref_net(tr.x[:, :28].double())
    pred = tr.base.double() + torch.cat([float(rob.ds) * out[:, :19], out[:, 19:]], dim=1)
    tgt = tr.target_rows[: tr.Q].double()
    K, steps = 4, T - 1
    e_pred = quaternion_to_euler(pred[:, 3:7].t().float()).double()  # the reference's loss runs this part in fp32
    e_tgt = quaternion_to_euler(tgt[:, 3:7].t().float()).double()
    # per window step the reference sums four nn.MSELoss (mean) terms over [rows, K] blocks, then sums the steps of
    # all trajectories and divides by 29 (physics_train.py:345-352, :266-267 / :396)
    total = (((pred[:, :3] - tgt[:, :3]) ** 2).sum() / (3 * K) + ((pred[:, 7:19] - tgt[:, 7:19]) ** 2).sum() / (12 * K)
             + ((e_pred - e_tgt) ** 2).sum() / (3 * K) + ((pred[:, 19:] - tgt[:, 19:]) ** 2).sum() / (6 * K)) / steps
    total.backward()
    assert abs(got_loss - float(total)) < 1e-4 * abs(float(total))
    for a, p in zip(got_grads, ref_net.parameters()):
        assert rel_l2(a.cpu().numpy(), p.grad.cpu().numpy()) < 2e-4
    # update: Adam(lr 1e-2) + clamp of every weight matrix
    params32 = [nn.Parameter(w.clone()) for w in w0]
    for p, gr in zip(params32, got_grads):
        p.grad = gr.clone()
    opt = torch.optim.Adam(params32, lr=1e-2)
    opt.step()
    with torch.no_grad():
        for k in (0, 2):
            params32[k].clamp_(min=0)
    tr.apply_update()
    torch.cuda.synchronize()
    for a, b in zip(rob.nn_models.parameters(), params32):
        # the first Adam step moves every weight by ~lr = 1e-2 against values of ~1e-2: compare absolutely (same bar
        # as test_cfg3_full_size_training_step)
        assert float((a.detach() - b.detach()).abs().max()) < 1e-6
    assert float(rob.nn_models[0].weight.min()) >= 0 and float(rob.nn_models[2].weight.min()) >= 0
    # the SAME epoch through kr_train_epoch (tr.step(): mlp_fwd2_kernel + loss epilogue, mlp_bwd2_kernel, train_tail_kernel -
    # what bench.py's cfg4_shard_epoch and train_dp legs time) from the same initial weights, against fp64 torch +
    # torch.optim.Adam above
    rob2 = CosseratRodTorch(DEV, 512)
    setup_robot(rob2, "damping")
    rob2.N = N
    rob2.compute_intermediate_terms()
    with torch.no_grad():
        for a, b in zip(rob2.nn_models.parameters(), w0):
            a.copy_(b)
    tr2 = KnodeTrainer(rob2, traj, controls, kp)
    l_epoch = tr2.step()
    assert tr2.fused_epoch
    assert abs(l_epoch - float(total)) < 1e-4 * abs(float(total))
    off = 0
    for k, (a, b, gref) in enumerate(zip(rob2.nn_models.parameters(), params32, ref_net.parameters())):
        assert float((a.detach() - b.detach()).abs().max()) < 1e-6, k
        n = a.numel()
        g64 = gref.grad.reshape(-1)
        m = tr2.exp_avg[off:off + n].double()
        assert float((m - 0.1 * g64).norm() / (0.1 * g64).norm()) < 2e-4, k  # exp_avg after step 1 = (1 - beta1) g
        off += n
    # a second epoch runs on the updated weights and lowers or keeps the loss scale finite
    l2 = float(tr.step())
    assert np.isfinite(l2)


def _train_cmd_64():
    """64 trajectories on the command line of physics_train.py:37-50 ("types... args...")."""
    rng = np.random.default_rng(1236)
    types, cargs = [], []
    for m in range(64):
        if m % 2 == 0:
            types.append("sine")
            cargs.append(f"{rng.uniform(0.5, 3.0):.3f}")
        else:
            types.append("random")
            cargs.append(str(int(rng.integers(1, 10000))))
    return types + cargs


def test_cfg4_two_rank_loss_curve(torch_cuda, tmp_path):
    """train_knode.py (the loop of physics_train.py:306-417 on the fused kernels) on 64 trajectories: two ranks under
    torch.distributed.run (gloo rendezvous, both on the one GPU; 32 trajectories each, one flat-gradient all-reduce per
    epoch) follow the single-process loss curve over 21 epochs.  The two runs sum the same per-trajectory terms in a
    different order in fp32, so the curves agree to rounding, not bitwise."""
    import krod_checkpoint as kc
    env = dict(os.environ, PYTHONPATH=PKG, KR_DIST_BACKEND="gloo")
    script = os.path.join(PKG, "train_knode.py")
    common = _train_cmd_64() + ["--fast", "--mod", "damping", "--epochs", "21", "--layers", "512", "--no-eval"]

    def run(prefix, save):
        out = subprocess.run(prefix + [script] + common + ["--save", save], env=env, capture_output=True, text=True,
                             timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        assert "Total number of trajectories:  64" in out.stdout
        return np.asarray([float(v) for v in kc.load_checkpoint(save, DEV)["loss"]])

    one = run([sys.executable], str(tmp_path / "one.pth"))
    port = str(29900 + os.getpid() % 90)
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", port], str(tmp_path / "two.pth"))
    assert one.shape == two.shape == (21,)
    assert np.all(np.isfinite(one)) and one[-1] < one[0]
    assert np.max(np.abs(one - two) / np.abs(one)) < 2e-5, np.max(np.abs(one - two) / np.abs(one))
    assert abs(one[0] - two[0]) <= 2e-6 * abs(one[0])  # epoch 0: same weights, only the summation order differs


# ---------------------------------------------------------------------------
# boundary
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["mid_N40_default", "mid_N100_None"])
def test_rk4_residual_honours_callers_midpoints(torch_cuda, monkeypatch, tag):
    """getResidualRK4(G, y, z, yh, yh_int, zh, zh_int) with yh_int / zh_int that are NOT the interpolation of yh, zh:
    the reference's stages 2 and 3 read them as passed (cosserat_ode.py:225,233-234)."""
    set_mode_env(monkeypatch, "single")
    g = load_golden("round3")
    N = int(tag.split("_")[1][1:])
    mod = tag.split("_")[2]
    r = make_robot("default" if mod == "default" else None, N)
    y0, z0, yp, zp = g[f"{tag}_y"], g[f"{tag}_z"], g[f"{tag}_yp"], g[f"{tag}_zp"]
    yh = r.c1 * y0 + r.c2 * yp
    zh = r.c1 * z0 + r.c2 * zp
    r.tendon_tensions = g[f"{tag}_tens"]
    for k, G in enumerate(g[f"{tag}_G"]):
        y, z = y0.copy(), z0.copy()
        res = r.getResidualRK4(G, y, z, yh, g[f"{tag}_yh_int"], zh, g[f"{tag}_zh_int"])
        assert rel_l2(y, g[f"{tag}_yout"][k]) < 1e-10
        assert rel_l2(z, g[f"{tag}_zout"][k]) < 1e-10
        assert np.allclose(res, g[f"{tag}_r"][k], rtol=1e-8, atol=1e-10 * np.abs(y[7:13]).max())
        assert np.array_equal(z[:, -1], z0[:, -1])
        # and the interpolated midpoints give a different sweep (the argument is not ignored)
        y2, z2 = y0.copy(), z0.copy()
        r.getResidualRK4(G, y2, z2, yh, 0.5 * (yh[:, :-1] + yh[:, 1:]), zh, 0.5 * (zh[:, :-1] + zh[:, 1:]))
        assert rel_l2(y2, y) > 1e-6
    # Euler ignores them, like the reference (cosserat_ode.py:188-213)
    y, z = y0.copy(), z0.copy()
    a = r.getResidualEuler(g[f"{tag}_G"][0], y, z, yh, g[f"{tag}_yh_int"], zh, g[f"{tag}_zh_int"])
    y, z = y0.copy(), z0.copy()
    b = r.getResidualEuler(g[f"{tag}_G"][0], y, z, yh, None, zh, None)
    assert np.array_equal(a, b)


def test_simulate_leaves_last_control_on_the_robot(torch_cuda):
    """knode.py:71: after simulate() the robot holds the LAST control (the one whose solve is dropped), so a caller's
    following getResidualEuler sees it."""
    from knode import simulate
    g = load_golden("sim_cfg1")
    r = make_robot(None, 20)
    assert r.tendon_tensions is None
    ctl = g["ctl"][:6].copy()
    ctl[-1] = [7.0, 5.5, 5.0, 6.5]
    traj = simulate(r, ctl)
    assert np.array_equal(np.asarray(r.tendon_tensions), ctl[-1]) and np.asarray(r.tendon_tensions).dtype == np.float64
    # ... and it is usable: one residual sweep at the last entry's base wrench
    y, z = traj[-1, :19].copy(), traj[-1, 19:25].copy()
    res = r.getResidualEuler(traj[-1, 7:13, 0].copy(), y, z, traj[-1, 25:44], None, traj[-1, 44:50], None)
    assert res.shape == (6,) and np.all(np.isfinite(res))


# ---------------------------------------------------------------------------
# multi-rank entry points
# ---------------------------------------------------------------------------
def test_bench_two_ranks(torch_cuda):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per rank), rehearsed with the
    gloo backend and both ranks on the one GPU: one JSON line, whole-job value, every rod-step converged."""
    env = dict(os.environ, KR_BENCH_BACKEND="gloo")
    port = str(29800 + os.getpid() % 90)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20",
           "--warmup", "5", "--no-cpu"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 20 and rec["warmup"] == 5 and rec["scaling"] == "weak"
    assert rec["config"]["rods_per_gpu"] == 1024 and rec["config"]["unconverged_rod_steps"] == 0
    # whole-job aggregate: all ranks' rod-steps over the slowest rank's time (ms_per_step is rounded to 1e-4 ms)
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 1024 * 20 / (rec["ms_per_step"] * 20e-3)) < 2e-2 * rec["value"]
    assert "roofline" in rec and rec["unit"] == "rod-steps/s"
    assert rec["timed_chunks"]["n"] == 5 and len(rec["timed_chunks"]["wall_ms"]) == 5
    # round 4: the driver's N > 1 command also runs BASELINE cfg4 data parallel - 4096 trajectories over the ranks, ONE
    # all-reduce of the flat gradient + loss buffer per epoch (gloo here, RCCL under the nccl backend), 50 epochs - and
    # rank 0 repeats the global batch alone: same loss curve
    dp = rec["extra"]["train_dp"]
    assert "error" not in dp, dp
    assert dp["ranks_seen"] == 2 and dp["trajectories"] == 4096 and dp["trajectories_per_rank"] == 2048
    assert dp["epochs"] == 50 and dp["floats"] == 28 * 512 + 512 + 512 * 25 + 25 + 1
    assert dp["allreduce_us"] is not None and dp["allreduce_us"] > 0 and dp["data_unconverged"] == 0
    assert np.isfinite(dp["loss_first"]) and dp["loss_last"] < dp["loss_first"]
    chk = dp["single_rank_check"]
    assert chk["ok"] and chk["rel_dev_first"] < 2e-5, chk
    # round 5: the leg explains itself - all-reduce min / median / max, the epoch with and without the collective
    ar = dp["allreduce_us_stats"]
    assert ar["n"] == 50 and 0 < ar["min"] <= ar["median"] <= ar["max"]
    assert dp["epoch_us_with_allreduce"] > 0 and dp["epoch_us_without_allreduce"] > 0
    # strong scaling beside the default weak mode: 1024 rods split over the two ranks
    st = rec["extra"]["strong_scaling"]
    assert st["rods_per_gpu"] == 512 and st["unconverged"] == 0 and st["value"] > 0
