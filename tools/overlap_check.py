#!/usr/bin/env python3
"""Overlapped persistent kernel (kr_mso_impl.hpp) against the one-wavefront persistent kernel on the bench workload:
same tips, same status; time per step of both.  python tools/overlap_check.py [B] [T] [N] [dtype] [kind]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from cosserat_ode import CosseratRod
from knode import setup_robot

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
dt = torch.float64 if (len(sys.argv) <= 4 or sys.argv[4] == "f64") else torch.float32
kind = sys.argv[5] if len(sys.argv) > 5 else "sine"
dev = "cuda:0"


def controls():
    if kind == "sine":
        return bench.rank_controls(B, 1, 0, T, 0.05)
    rng = np.random.default_rng(5)
    if kind == "random":  # fresh random tensions every step (physics_controls 'random')
        return 5.0 + 5.0 * rng.uniform(size=(B, T, 4))
    if kind == "step":
        c = np.full((B, T, 4), 5.0)
        c[:, T // 3:, 0] += 1.0
        c[:, T // 3:, 3] += 1.0
        return c
    raise SystemExit(kind)


def run(overlap, maxit=0):
    r = CosseratRod(use_fsolve=True)
    setup_robot(r)
    r.N = N
    r.compute_intermediate_terms()
    h = r._native()
    h.set_option("overlap", overlap)
    h.set_option("waves_per_rod", int(os.environ.get("KR_OC_WAVES", "1")))
    ctl = torch.as_tensor(controls(), device=dev).to(dt).contiguous()
    st = h.new_state(B, dt, n_slots=3)
    G = torch.zeros((B, 6), dtype=dt, device=dev)
    tip = torch.empty((B, T, 3), dtype=dt, device=dev)
    status = torch.full((B, T), -1, dtype=torch.int32, device=dev)
    best = 1e9
    for rep in range(3):
        h.init_straight(st[0])
        G.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h.simulate(ctl, st, G, ring=True, tip=tip, status=status, maxit=maxit)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("   waves per rod:", h.get_option("last_waves_per_rod"), "path", h.get_option("last_sim_path"))
    return tip.double().cpu().numpy(), status.cpu().numpy(), best, h.get_option("last_overlap"), st[T % 3].double().cpu().numpy()


t1, s1, d1, o1, f1 = run(1)
t0, s0, d0, o0, f0 = run(0)
print(f"B={B} T={T} N={N} {dt} {kind}: overlapped ran={o1} {d1 / T * 1e6:.2f} us/step ({B * T / d1 / 1e6:.2f} M rod-steps/s); "
      f"classic ran_overlap={o0} {d0 / T * 1e6:.2f} us/step ({B * T / d0 / 1e6:.2f} M)")
print("status nonzero: overlapped", int((s1 != 0).sum()), "classic", int((s0 != 0).sum()))
err = np.linalg.norm((t1 - t0).reshape(B, -1), axis=1) / np.linalg.norm(t0.reshape(B, -1), axis=1)
print("tip rel L2 overlapped vs classic: max %.3e" % err.max(), " final state rel diff %.3e" % (np.abs(f1 - f0).max() / np.abs(f0).max()))
