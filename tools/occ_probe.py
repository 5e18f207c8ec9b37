import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from cosserat_ode import CosseratRod
from knode import setup_robot
dev = "cuda:0"
for N in (100, 64):
    r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
    h = r._native(); h.set_option("waves_per_rod", 1); h.set_option("keep_predictor", 1)
    for dt in (torch.float32, torch.float64):
        for B in (1024, 2048, 4096):
            ctl = torch.as_tensor(bench.rank_controls(B, 1, 0, 60 + 300, r.del_t), device=dev).to(dt).contiguous()
            best = 1e9
            for rep in range(2):
                st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
                h.simulate(ctl[:, :60].contiguous(), st, G, ring=True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                h.simulate(ctl[:, 60:].contiguous(), st, G, ring=True, prev_init=st[2].clone())
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            print(f"N={N} {str(dt)[6:]:8s} B={B:5d}: {best/300*1e6:7.1f} us/step  {B*300/best/1e6:6.2f} M rod-steps/s  path {h.get_option('last_sim_path')}", flush=True)
