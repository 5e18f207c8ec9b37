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
M, T, N, kp, layers = 1024, 64, 100, [22, 67, 99], [64, 64]
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
ctl = orc.batch_sine_controls(M, T, r.del_t, 1236)
out = simulate_batch(r, ctl, dtype="f32")
traj = torch.as_tensor(out["traj"][:, :T], device=dev).float().contiguous()
controls = torch.as_tensor(ctl, device=dev).float().contiguous()
rob = CosseratRodTorch(dev, 64); setup_robot(rob, "damping"); rob.N = N; rob.compute_intermediate_terms()
mods = [nn.Linear(28, 64), nn.ELU(), nn.Linear(64, 64), nn.ELU(), nn.Linear(64, 25)]
rob.nn_models = nn.ModuleList(mods).to(dev)
tr = KnodeTrainer(rob, traj, controls, kp)
for _ in range(6): tr.step(sync_loss=False)
torch.cuda.synchronize()
