#!/usr/bin/env python3
"""KNODE training driver with the command line, loop shape and printed lines of the reference's
``physics_train.py`` (argument parser :37-50, data generation :98-134, epoch loop :209-304 / :306-408,
evaluation :136-167, checkpoint :284-288), running on the MI355X kernels:

  * reference trajectories: ``knode.simulate(robot_reference, controls)`` (persistent multiple-shooting kernel),
  * epoch: ``krod_train.KnodeTrainer.step`` - fused MLP forward / loss / backward / Adam + clamp, every
    (trajectory, window step, key point) row in one batch,
  * evaluation every 50 epochs: closed-loop rollout with the live weights (``krod_eval.evaluate``), FastDTW of the
    tip path against the validation reference (radius 1, L1: what the reference's ``fastdtw`` call computes, restated),
  * checkpoint: ``torch.save({'robot', 'dtw', 'loss', 'optim'})`` - readable by the reference; ``optim`` is an
    Adam ``state_dict`` (step, exp_avg, exp_avg_sq per parameter + param_groups), ``--resume`` continues from it.

The printed lines ``Epoch {n} of {epochs}`` and ``Total loss: {x}, lr {[..]}`` are the ones
``physics_multitrain.py:113-121`` parses.  Under ``torch.distributed.run`` the trajectories are sharded over the
ranks (data parallel, one all-reduce of the flat gradient buffer per epoch); rank 0 prints, evaluates and saves.

    python train_knode.py sine 2 --fast --epochs 200
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_knode.py sine random 2 7 --fast
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from cosserat_ode import CosseratRod  # noqa: E402
from cosserat_ode_torch import CosseratRodTorch  # noqa: E402
from knode import setup_robot, simulate  # noqa: E402
from physics_controls import calc_controls  # noqa: E402
import krod_eval  # noqa: E402
from krod_train import KnodeTrainer, shard_range  # noqa: E402

train_len = 30   # physics_train.py:33-35
eval_len = 100


def split_list(a):
    half = len(a) // 2
    return a[:half], a[half:]


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train KNODE (MI355X backend).")
    ap.add_argument("--eval", action=argparse.BooleanOptionalAction, default=True)
    ap.add_argument("--mod", type=str, default=None)
    ap.add_argument("control_type_arg", nargs="+", type=str, help='trajectories to train on, e.g. "sine 2" or "sine random 2 7"')
    ap.add_argument("--epochs", type=int, default=2000)
    ap.add_argument("--weight_decay", type=float, default=0)
    ap.add_argument("--noise_traj", type=float, default=0)
    ap.add_argument("--noise_controls", type=float, default=0)
    ap.add_argument("--layers", type=float, default=512)
    ap.add_argument("--validation", type=str, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--fast", action=argparse.BooleanOptionalAction, default=False)
    ap.add_argument("--save", type=str, default=None, help="checkpoint path (default: saved_models/<reference naming>)")
    ap.add_argument("--resume", type=str, default=None,
                    help="checkpoint to continue from (RESUME_TRAINING of physics_train.py:27,186-188,202-204)")
    args = ap.parse_args(argv)

    control_type, control_arg = split_list(args.control_type_arg)
    if len(control_type) != len(control_arg):
        raise Exception("Different number of control_type and control_arg")
    control_arg = [float(i) for i in control_arg]
    validation = args.validation or "sine 1.25"
    validation_type, validation_arg = validation.split(" ")
    validation_arg = float(validation_arg)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # KR_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the data-parallel path on a 1-GPU box)
    backend = os.environ.get("KR_DIST_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
    say = print if rank == 0 else (lambda *a, **k: None)

    data_short = f'physics_{"-".join(control_type)}_{"-".join(map(str, control_arg))}'.replace(".", "_")
    save_path = args.save or f"saved_models/{data_short}_{args.mod}_trainlen_{train_len}_{args.epochs}_epoch_{args.seed}.pth"
    say(save_path)

    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    robot_reference = CosseratRod(use_fsolve=True, device=dev_index)   # generates the data (true parameters)
    setup_robot(robot_reference)
    robot_eval = CosseratRod(use_fsolve=True, device=dev_index)        # imperfect model + MLP, for evaluation
    setup_robot(robot_eval, args.mod)
    robot = CosseratRodTorch(device, args.layers)                       # imperfect model + trainable MLP
    setup_robot(robot, args.mod)

    resume = None
    if args.resume:
        import krod_checkpoint as kc
        resume = kc.load_checkpoint(args.resume, device)
        robot.nn_models.load_state_dict(resume["robot"].nn_models.state_dict())

    # training data (physics_train.py:98-134): one reference trajectory per control specification
    trajs, ctls = [], []
    for ct, ca in zip(control_type, control_arg):
        controls = np.array(calc_controls(ct, ca, robot_reference.del_t, train_len))
        traj = simulate(robot_reference, controls)[:, :25]
        trajs.append(traj)
        ctls.append(controls)
    traj_t = torch.tensor(np.array(trajs), dtype=torch.float32, device=device)
    ctl_t = torch.tensor(np.array(ctls), dtype=torch.float32, device=device)
    traj_t = traj_t + torch.randn_like(traj_t) * args.noise_traj
    ctl_t = ctl_t + torch.randn_like(ctl_t) * args.noise_controls
    say("Total number of trajectories: ", len(trajs))
    # more ranks than trajectories: the surplus ranks hold an EMPTY shard and contribute zeros to the summed
    # gradient and loss (KnodeTrainer and the C API accept Q = 0), so the result equals the single-process one
    lo, hi = shard_range(len(trajs), rank, world)
    key_pt_idx = [3, 5, 7, 9] if args.fast else [2, 6, 9]   # physics_train.py:312 / :216-220
    trainer = KnodeTrainer(robot, traj_t[lo:hi], ctl_t[lo:hi], key_pt_idx, weight_decay=args.weight_decay)
    if resume is not None and resume.get("optim"):
        trainer.load_optimizer_state_dict(resume["optim"])

    validation_controls = np.array(calc_controls(validation_type, validation_arg, robot_reference.del_t, eval_len))
    validation_reference = simulate(robot_reference, validation_controls)[:, :25] if (args.eval and rank == 0) else None

    loss_arr, dtw_arr = [], []
    best = (float("inf"), None)
    first_epoch = trainer.scheduler.steps if trainer.device_plateau else 0  # (> 0 after --resume)
    for epoch in range(args.epochs):
        # no host round trip inside an epoch: loss, Adam, plateau schedule and clamp all advance on the device; the
        # loss is read back where the reference prints it (every 10 epochs, physics_train.py:270-272) and at the end
        printing = epoch % 10 == 0
        loss = trainer.step(sync_loss=printing or not trainer.device_plateau)
        if not trainer.device_plateau:
            loss_arr.append(loss)
        if printing:
            say(f"Epoch {epoch} of {args.epochs}")
            say(f"Total loss: {loss}, lr {trainer.scheduler.get_last_lr()}")
        if epoch % 50 == 0 and args.eval and rank == 0:
            dtw, _ = krod_eval.evaluate(robot_eval, robot if epoch != 0 else None, validation_controls,
                                        validation_reference, eval_len)
            dtw_arr.append([dtw])
            say("Validation DTW Distance XYZ", dtw)
            if dtw < best[0]:
                best = (dtw, {k: v.detach().clone() for k, v in robot.nn_models.state_dict().items()})
    if trainer.device_plateau:
        loss_arr = trainer.losses()[first_epoch:]
    if rank == 0:
        os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
        if best[1] is not None:
            robot.nn_models.load_state_dict(best[1])   # physics_train.py:410-417 keeps the best-DTW snapshot
        # 'optim': torch.optim.Adam's state_dict layout (physics_train.py:284-288), filled from the fused optimizer
        torch.save({"robot": robot, "dtw": dtw_arr, "loss": loss_arr, "optim": trainer.optimizer_state_dict()}, save_path)
        say("saved", save_path)
    if world > 1:
        dist.destroy_process_group()
    return loss_arr, dtw_arr


if __name__ == "__main__":
    main()
