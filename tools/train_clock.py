#!/usr/bin/env python3
"""GPU box: does the training epoch get faster under sustained load (shader clock ramp)?  cfg3 / cfg4 epochs back to back,
median HIP-event time per window of 500 epochs.    python tools/train_clock.py [cfg3|cfg4] [epochs]"""
import os, sys, statistics, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import torch
import bench_legs as bl
from krod_train import KnodeTrainer
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
M, T, N, key, layers = (1024, 64, 100, [22, 67, 99], [64, 64]) if cfg == "cfg3" else (512, 30, 10, [3, 5, 7, 9], [512])
dev = "cuda:0"
rr = bl.make_robot(N, 0)
ctl = bl.sine_controls(M, T, rr.del_t, 1236)
traj, bad = bl.device_trajectories(torch, rr, ctl)
rob, sizes = bl.torch_rod(torch, dev, N, layers)
tr = KnodeTrainer(rob, traj, torch.as_tensor(ctl, device=dev).float().contiguous(), key, keep_pred=False)
torch.cuda.synchronize()
W = int(sys.argv[3]) if len(sys.argv) > 3 else 500
t_all = time.perf_counter()
for w in range(E // W):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(W)]
    t0 = time.perf_counter()
    for a, b in evs:
        a.record(); tr.step(sync_loss=False); b.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / W * 1e6
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    print(f"{cfg} epochs {w * W:5d}..{(w + 1) * W - 1:5d}: median {statistics.median(us):7.1f} us  min {us[0]:7.1f}  wall per epoch {wall:7.1f} us  (t = {time.perf_counter() - t_all:.2f} s)", flush=True)
