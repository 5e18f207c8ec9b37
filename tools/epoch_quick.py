#!/usr/bin/env python3
"""GPU box: kr_train_epoch against the three separate calls - same losses / parameters, epoch time of both.
    python tools/epoch_quick.py [cfg3|cfg4]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench_legs as bl
from krod_train import KnodeTrainer

def run(cfg, fused, epochs=20):
    M, T, N, key, layers = (1024, 64, 100, [22, 67, 99], [64, 64]) if cfg == "cfg3" else (512, 30, 10, [3, 5, 7, 9], [512])
    dev = "cuda:0"
    rr = bl.make_robot(N, 0)
    ctl = bl.sine_controls(M, T, rr.del_t, 1236)
    traj, bad = bl.device_trajectories(torch, rr, ctl)
    rob, sizes = bl.torch_rod(torch, dev, N, layers)
    tr = KnodeTrainer(rob, traj, torch.as_tensor(ctl, device=dev).float().contiguous(), key, keep_pred=False)
    tr.fused_epoch = fused
    for _ in range(3):
        tr.step(sync_loss=False)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(epochs)]
    for a, b in evs:
        a.record(); tr.step(sync_loss=False); b.record()
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return statistics.median(us), np.array(tr.losses()), tr.flat_p.detach().cpu().numpy().copy(), tr.Q, tr.fused_epoch

for cfg in sys.argv[1:] or ["cfg3", "cfg4"]:
    t0, l0, p0, Q, _ = run(cfg, False)
    t1, l1, p1, Q, still = run(cfg, True)
    print(f"{cfg}: Q={Q} separate {t0:.1f} us, kr_train_epoch {t1:.1f} us (served {still}); loss rel diff max "
          f"{np.max(np.abs(l0 - l1) / np.abs(l0)):.2e}, params max abs diff {np.max(np.abs(p0 - p1)):.2e} (|p| max {np.abs(p0).max():.2f}); "
          f"loss first/last {l1[0]:.6g} {l1[-1]:.6g}", flush=True)
