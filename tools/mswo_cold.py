#!/usr/bin/env python3
"""GPU box: T steps from the straight rod in ONE call (no warm-up of the rod, no predictor hand-over; what knode.simulate does),
overlapped several-wavefront kernel against the plain form; sine tensions of the bench.  Best of 3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench_legs as bl
dev = "cuda:0"
for N, B in ((100, 256), (400, 512), (100, 128)):
    for T in (64, 200, 600):
        r = bl.make_robot(N, 0); h = r._native()
        c = torch.as_tensor(bl.sine_controls(B, T, r.del_t, 1234), device=dev).contiguous()
        line = f"N={N} B={B} T={T}:"
        for ov in (0, 1):
            h.set_option("msw_overlap", ov)
            best = 1e9
            for rep in range(4):
                st = h.new_state(B, torch.float64, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=torch.float64, device=dev)
                status = torch.zeros((B, T), dtype=torch.int32, device=dev)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                h.simulate(c, st, G, ring=True, status=status)
                torch.cuda.synchronize(); el = time.perf_counter() - t0
                if rep: best = min(best, el)
            line += f"  overlap={ov} (ran {h.get_option('last_overlap')}): {best / T * 1e6:6.1f} us/step bad {int((status != 0).sum())}"
        print(line, flush=True)
