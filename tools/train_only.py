#!/usr/bin/env python3
"""Dev tool (GPU box): the cfg3 training epoch alone, for profiling (rocprofv3 ... -- python3 tools/train_only.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, torch.nn as nn
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from cosserat_ode_torch import CosseratRodTorch
from knode import setup_robot, simulate_batch
from krod_train import KnodeTrainer
dev = "cuda:0"
M, T, N, kp, layers = int(os.environ.get("KR_TRAIN_M", "1024")), 64, 100, [22, 67, 99], [64, 64]
if len(sys.argv) > 1 and sys.argv[1] == "cfg4":
    M, T, N, kp, layers = 512, 30, 100, [33, 55, 77, 99], [512]
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
ctl = orc.batch_sine_controls(M, T, r.del_t, 1236)
out = simulate_batch(r, ctl, dtype="f32")
traj = torch.as_tensor(out["traj"][:, :T], device=dev).float().contiguous()
controls = torch.as_tensor(ctl, device=dev).float().contiguous()
rob = CosseratRodTorch(dev, layers[0]); setup_robot(rob, "damping"); rob.N = N; rob.compute_intermediate_terms()
if len(layers) == 2:
    mods = [nn.Linear(28, layers[0]), nn.ELU(), nn.Linear(layers[0], layers[1]), nn.ELU(), nn.Linear(layers[1], 25)]
    for m in mods:
        if isinstance(m, nn.Linear):
            rob.non_negative_normal_init(m, 0.01, 0.01); nn.init.normal_(m.bias, 0.0, 0.01)
    rob.nn_models = nn.ModuleList(mods).to(dev)
tr = KnodeTrainer(rob, traj, controls, kp, keep_pred=False)
for _ in range(20): tr.step(sync_loss=False)
torch.cuda.synchronize()
print("done", tr.Q, "losses", tr.losses()[0], tr.losses()[-1], "lib", __import__("krod_native").LIB_PATH)
