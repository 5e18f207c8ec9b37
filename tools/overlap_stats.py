#!/usr/bin/env python3
"""Dev tool (GPU box): in-kernel counters of the overlapped persistent kernel (kr_mso_impl.hpp).
Needs the diagnostic library: make dbg; KR_LIB_PATH=knode-cosserat_amd/lib/dbg/libknode_rod.so
python tools/overlap_stats.py [B] [T] [warm]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, krod_native as kn
from cosserat_ode import CosseratRod
from knode import setup_robot
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
WARM = int(sys.argv[3]) if len(sys.argv) > 3 else 30
KIND = sys.argv[4] if len(sys.argv) > 4 else "sine"
N = 100
dev = "cuda:0"; dt = torch.float64
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native(); h.set_option("waves_per_rod", int(os.environ.get("KR_OC_WAVES", "1"))); h.set_option("keep_predictor", 1)
dbg = torch.zeros((B, 24), dtype=torch.int64, device=dev)
kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(dbg)))
if KIND == "sine":
    ctl_np = bench.rank_controls(B, 1, 0, WARM + T, r.del_t)
elif KIND == "random":  # fresh random tensions every step (physics_controls.py 'random')
    ctl_np = 5.0 + 5.0 * np.random.default_rng(5).uniform(size=(B, WARM + T, 4))
else:  # step
    ctl_np = np.full((B, WARM + T, 4), 5.0); ctl_np[:, WARM + T // 3:, 0] += 1.0; ctl_np[:, WARM + T // 3:, 3] += 1.0
ctl_all = torch.as_tensor(ctl_np, device=dev).contiguous()
st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
status = torch.full((B, T), -1, dtype=torch.int32, device=dev)
h.simulate(ctl_all[:, :WARM].contiguous(), st, G, ring=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
h.simulate(ctl_all[:, WARM:].contiguous(), st, G, ring=True, prev_init=st[(WARM - 1) % 3], status=status)
torch.cuda.synchronize(); el = time.perf_counter() - t0
d = dbg.cpu().numpy().astype(np.float64)
print(f"[{KIND}] overlap ran: {h.get_option('last_overlap')}  wall {el / T * 1e6:.2f} us/step  unconverged {int((status != 0).sum())}")
names = ["total", "t_sweep", "t_alg", "t_pred", "sweeps", "merged", "quick", "chord", "rejects", "retries", "rebuilds", "resume_at", "t_verdict", "t_cond(handover+chain+solve+updY)", "t_finish", "t_pred_update(copy+update)"]
for k, n in enumerate(names):
    print(f"  {n:10s} per step: mean {d[:, k].mean() / T:10.3f}  max {d[:, k].max() / T:10.3f}")
print("  ticks per sweep:", d[:, 1].sum() / d[:, 4].sum(), " per condensation:", d[:, 2].sum() / d[:, 4].sum(), " tick rate MHz", d[:, 0].max() / el / 1e6)
ex = ["pred: errors of the predictors in use", "pred: refit", "pred: choice of the order", "verdict: end states + last record", "verdict: residual norm",
      "cond: columns through Es + residual norm", "cond: chain", "hand-over: copy of the unknowns"]
for k, n in enumerate(ex):
    print(f"  {n:44s} per step: mean {d[:, 16 + k].mean() / T:9.1f}")
