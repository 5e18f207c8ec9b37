#!/usr/bin/env python3
"""GPU box: random shapes through kr_train_epoch against the three separate calls (same data, 4 epochs each).
    python tools/soak_epoch.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch, torch.nn as nn
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from cosserat_ode_torch import CosseratRodTorch
from knode import setup_robot, simulate_batch
from krod_train import KnodeTrainer
dev = "cuda:0"
N = 16
ACTS = {"elu": nn.ELU, "tanh": nn.Tanh, "softplus": nn.Softplus, "relu": nn.ReLU}


def run(cases, seed, verbose=True):
  """Number of mismatching cases."""
  rng = np.random.default_rng(seed)
  r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
  bad = 0
  for case in range(cases):
      M = int(rng.choice([1, 2, 3, 7, 24, 64, 300]))
      T = int(rng.choice([2, 3, 9, 30]))
      K = int(rng.integers(1, 6))
      kp = sorted(rng.choice(np.arange(1, N), size=K, replace=False).tolist())
      if rng.random() < 0.5:
          layers = [int(rng.choice([8, 40, 64, 100, 256, 512]))]
      else:
          layers = [int(rng.choice([8, 33, 64])), int(rng.choice([16, 48, 64]))]
      act = str(rng.choice(list(ACTS)))
      ctl = orc.batch_sine_controls(M, T, r.del_t, 100 + case)
      traj = torch.as_tensor(simulate_batch(r, ctl, dtype="f32")["traj"][:, :T], device=dev).float().contiguous()
      out = []
      for fused in (True, False):
          rob = CosseratRodTorch(dev, layers[0]); setup_robot(rob, "damping"); rob.N = N; rob.compute_intermediate_terms()
          torch.manual_seed(case)
          sizes = [28] + layers + [25]
          mods = []
          for a, b in zip(sizes[:-1], sizes[1:]):
              mods += [nn.Linear(a, b), ACTS[act]()]
          mods = mods[:-1]
          for m in mods:
              if isinstance(m, nn.Linear):
                  rob.non_negative_normal_init(m, 0.01, 0.01); nn.init.normal_(m.bias, 0.0, 0.01)
          rob.nn_models = nn.ModuleList(mods).to(dev)
          tr = KnodeTrainer(rob, traj, torch.as_tensor(ctl, device=dev).float().contiguous(), kp, keep_pred=False)
          tr.fused_epoch = fused
          for _ in range(4):
              tr.step(sync_loss=False)
          torch.cuda.synchronize()
          out.append((np.array(tr.losses()), tr.flat_p.cpu().numpy().copy(), tr.fused_epoch, tr.Q))
      (la, pa, served, Q), (lb, pb, _, _) = out
      dl = float(np.max(np.abs(la - lb) / np.maximum(np.abs(lb), 1e-30)))
      dp = float(np.max(np.abs(pa - pb)))
      ok = np.all(np.isfinite(la)) and dl < 2e-5 and dp < 5e-6 * max(1.0, float(np.abs(pb).max()))
      bad += not ok
      if verbose: print(f"case {case:2d}: M={M:3d} T={T:2d} K={K} Q={Q:6d} 28-{'-'.join(map(str, layers))}-25 {act:8s} epoch call served {served}: "
            f"loss rel diff {dl:.1e}, params diff {dp:.1e} {'ok' if ok else 'MISMATCH'}", flush=True)
  return bad


if __name__ == "__main__":
    n_bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("mismatches:", n_bad)
    sys.exit(1 if n_bad else 0)
