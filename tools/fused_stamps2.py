#!/usr/bin/env python3
"""GPU box: phase stamps of the two-layer forward kernel (cfg4 shard: 28 -> 512 -> 25, Q = 59 392).
    KR_LIB_PATH=knode-cosserat_amd/lib/dbg/libknode_rod_fused.so python tools/fused_stamps2.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench_legs as bl
import krod_native as kn
from krod_train import KnodeTrainer
M, T, N, key, layers = 512, 30, 10, [3, 5, 7, 9], [512]
dev = "cuda:0"
rr = bl.make_robot(N, 0)
ctl = bl.sine_controls(M, T, rr.del_t, 1236)
traj, bad = bl.device_trajectories(torch, rr, ctl)
rob, sizes = bl.torch_rod(torch, dev, N, layers)
tr = KnodeTrainer(rob, traj, torch.as_tensor(ctl, device=dev).float().contiguous(), key, keep_pred=False)
for _ in range(3):
    tr.step(sync_loss=False)
torch.cuda.synchronize()
dbg = torch.zeros(3 * 4096 * 12, dtype=torch.int64, device=dev)
kn.check(tr.h.lib.kr_debug_buffer(tr.h._h, kn._ptr(dbg)))
E = 10
for _ in range(E):
    tr.step(sync_loss=False)
torch.cuda.synchronize()
raw = dbg.cpu().numpy().astype(np.float64).reshape(3, 4096, 12) / E
d = raw[0].sum(0)
nblk = (tr.Q + 31) // 32
ph = ["wait x rows", "bias of a chunk (x8)", "layer 1 products (x8)", "activation (x8)", "layer 2 products (x8)", "-", "-", "outputs -> tile", "loss epilogue", "store rows"]
tot = d[:10].sum()
print(f"two-layer forward: {tot / nblk:9.0f} cycles per row block of one wavefront ({nblk} blocks)")
for i, p in enumerate(ph):
    print(f"    {p:34s} {d[i] / nblk:9.0f}  {100 * d[i] / tot:5.1f} %")
