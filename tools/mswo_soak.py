#!/usr/bin/env python3
"""GPU box: long runs of the overlapped several-wavefront kernel against the plain form at the bench's batch sizes
(cfg2: B = 256, N = 100, W = 4, tiles in the LDS; cfg5: B = 512, N = 400, W = 2, tiles read from the states), fp64 and fp32,
on sine tensions with random jumps: every rod-step converged, tips within the stopping tolerance of each other."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench_legs as bl
dev = "cuda:0"
bad = 0
for N, B, T in ((100, 256, 1500), (400, 512, 300)):
    for dt in (torch.float64, torch.float32):
        rng = np.random.default_rng(N + (1 if dt == torch.float32 else 0))
        r = bl.make_robot(N, 0); h = r._native()
        ctl = bl.sine_controls(B, T, r.del_t, 77)
        for b in range(B):  # three jumps per rod
            for t0 in rng.integers(1, T, size=3):
                ctl[b, t0:, rng.integers(0, 4)] += rng.uniform(-1.5, 1.5)
        c = torch.as_tensor(ctl, device=dev).to(dt).contiguous()
        res = []
        for ov in (0, 1):
            h.set_option("msw_overlap", ov)
            st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
            tip = torch.empty((B, T, 3), dtype=dt, device=dev); status = torch.zeros((B, T), dtype=torch.int32, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h.simulate(c, st, G, ring=True, tip=tip, status=status)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
            res.append((tip.double().cpu().numpy(), int((status != 0).sum()), el, h.get_option("last_overlap"), h.get_option("last_waves_per_rod")))
        a, b = res
        err = np.abs(a[0] - b[0]).max() / np.abs(a[0]).max()
        lim = 2e-4 if dt == torch.float32 else 2e-7
        ok = a[1] == 0 and b[1] == 0 and err < lim and b[3] == 1 and np.isfinite(b[0]).all()
        bad += 0 if ok else 1
        print(f"{'ok ' if ok else 'BAD'} N={N} B={B} T={T} {str(dt)[6:]}: W={b[4]} unconverged {a[1]} / {b[1]}, tips max rel diff {err:.2e}, "
              f"{a[2] / T * 1e6:.1f} -> {b[2] / T * 1e6:.1f} us per step from the straight rod", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
