#!/usr/bin/env python3
"""GPU box: where a wavefront of the fused training kernels spends its cycles, phase by phase (cfg3).
Needs the diagnostic library (make -C knode-cosserat_amd/csrc dbgf):
    KR_LIB_PATH=knode-cosserat_amd/lib/dbg/libknode_rod_fused.so python tools/fused_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench_legs as bl
import krod_native as kn
from krod_train import KnodeTrainer
M, T, N, key, layers = 1024, 64, 100, [22, 67, 99], [64, 64]
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
d = np.zeros(48); d[0:12] = raw[0].sum(0); d[16:28] = raw[1].sum(0); d[32:44] = raw[2].sum(0)
nblk = (tr.Q + 31) // 32
names = {
    "forward": (0, ["wait x rows", "bias 1", "layer 1 products", "act 1 + dump + bias 2", "layer 2 products", "act 2 + dump", "layer 3 products",
                    "outputs -> tile", "loss epilogue", "store rows", "(loop head: wait for x)", "(request next x)"]),
    "bwd3a": (16, ["stage dOUT^T", "A1, A2 -> tiles", "dW3 += dOUT^T A2", "d2 = W3^T dOUT", "act' + dump dZ2 + dZ2 -> tile", "dW2 += dZ2^T A1"]),
    "bwd3b": (32, ["stage X^T", "dZ2, A1 (-> tile)", "d1 = W2^T dZ2", "act' + dZ1 -> tile", "dW1 += dZ1^T X"]),
}
for k, (base, ph) in names.items():
    tot = d[base:base + len(ph)].sum()
    print(f"{k}: {tot / nblk:9.0f} cycles per row block of one wavefront ({nblk} blocks)")
    for i, p in enumerate(ph):
        print(f"    {p:34s} {d[base + i] / nblk:9.0f}  {100 * d[base + i] / tot:5.1f} %")
