#!/usr/bin/env python3
"""Dev tool (GPU box): the pipelined persistent kernel (kr_pipe_impl.hpp) against the one-wavefront persistent kernel:
same trajectories (to the stopping tolerance), all steps converged, time per step.   python tools/pipe_check.py [N B T]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from knode import setup_robot
N, B, T = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (100, 8, 30)
dev = "cuda:0"
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native()
h.set_option("waves_per_rod", 1)
for dt in (torch.float64, torch.float32):
    ctl = torch.as_tensor(orc.batch_sine_controls(B, T, r.del_t, 1235), device=dev).to(dt).contiguous()
    res = {}
    import krod_native as kn
    dbg = torch.zeros((B, 24), dtype=torch.int64, device=dev)
    for pipe in (0, 2):
        h.set_option("pipeline", pipe)
        kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(dbg) if pipe else None))
        best = 1e9
        for rep in range(3):
            st = h.new_state(B, dt, n_slots=T + 1); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
            status = torch.full((B, T), -1, dtype=torch.int32, device=dev); tip = torch.zeros((B, T, 3), dtype=dt, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h.simulate(ctl, st, G, status=status, tip=tip)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res[pipe] = (st.double().cpu().numpy(), G.double().cpu().numpy(), tip.double().cpu().numpy())
        print(f"{str(dt)[6:]:8s} pipeline={pipe} (ran pipelined: {h.get_option('last_pipelined')}, path {h.get_option('last_sim_path')}): {best/T*1e6:7.1f} us/step  "
              f"{B/(best/T)/1e6:6.2f} M rod-steps/s  status counts {np.bincount(status.cpu().numpy().ravel() + 1, minlength=4)[:4]} (index 0 = untouched)", flush=True)
    d = dbg.cpu().numpy().astype(np.float64)
    print(f"         solver ticks per step: total {d[:,0].mean()/T:.0f}  in solve {d[:,1].mean()/T:.0f}  waiting for verdicts {d[:,2].mean()/T:.0f}  redo {d[:,3].mean()/T:.3f}/step  sweeps {d[:,4].mean()/T:.2f}/step | "
          f"integrator: integrating {d[:,5].mean()/T:.0f}  waiting for publications {d[:,6].mean()/T:.0f}", flush=True)
    kn.check(h.lib.kr_debug_buffer(h._h, None))
    a, b = res[0], res[2]
    sc = np.abs(a[0]).max()
    print(f"         max |state diff| / scale {np.abs(a[0]-b[0]).max()/sc:.2e}   G {np.abs(a[1]-b[1]).max()/max(1,np.abs(a[1]).max()):.2e}   tip {np.abs(a[2]-b[2]).max()/np.abs(a[2]).max():.2e}   "
          f"slots 25.. of the pipelined states all zero: {float(np.abs(b[0][..., 25:]).max()) == 0.0}", flush=True)
