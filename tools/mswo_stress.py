#!/usr/bin/env python3
"""GPU box: the several-wavefront persistent kernel with overlapped steps (option msw_overlap, kr_mswo_impl.hpp) against the
same kernel family without the overlap on ROUGH inputs - fresh random tensions every step, random jumps, iteration caps,
presets, ring / trajectory, chunked calls with the predictor handed over.  Exits non-zero on a mismatch.
    python tools/mswo_stress.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
from cosserat_ode import CosseratRod
from knode import setup_robot
dev = "cuda:0"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
mods = [None, "noair", "nsw", "short", "damping", "dampstiff", "lengthstiff", "youngs"]

def controls(kind, B, T):
    if kind == "random":
        return 5.0 + 5.0 * rng.uniform(size=(B, T, 4))
    if kind == "jumps":
        c = np.full((B, T, 4), 5.0)
        for b in range(B):
            for t0 in sorted(rng.integers(0, T, size=3)):
                c[b, t0:, rng.integers(0, 4)] += rng.uniform(-2, 2)
        return c
    t = np.arange(T)[None, :, None] * 0.05
    return 5.0 + 3.0 * np.sin(2 * np.pi * rng.uniform(0.2, 2.0, size=(B, 1, 4)) * t + rng.uniform(0, 6, size=(B, 1, 4)))

def run(h, ctl, B, T, ring, chunks, overlap, maxit, dt=torch.float64):
    h.set_option("msw_overlap", overlap); h.set_option("keep_predictor", 1 if chunks > 1 else 0)
    c = torch.as_tensor(ctl, device=dev).to(dt).contiguous()
    tip = torch.empty((B, T, 3), dtype=dt, device=dev); status = torch.zeros((B, T), dtype=torch.int32, device=dev)
    G = torch.zeros((B, 6), dtype=dt, device=dev)
    ran = []
    if chunks == 1:
        st = h.new_state(B, dt, n_slots=3 if ring else T + 1); h.init_straight(st[0])
        h.simulate(c, st, G, ring=ring, tip=tip, status=status, maxit=maxit)
        ran.append(h.get_option("last_overlap"))
        last = st[T % 3 if ring else T].double().cpu().numpy()
    else:
        st = h.new_state(B, dt, n_slots=T + 1); h.init_straight(st[0])
        step = T // chunks; t0 = 0
        while t0 < T:
            n = min(step, T - t0)
            tp = torch.empty((B, n, 3), dtype=dt, device=dev); ss = torch.zeros((B, n), dtype=torch.int32, device=dev)
            h.simulate(c[:, t0:t0 + n].contiguous(), st[t0:t0 + n + 1], G, tip=tp, status=ss, maxit=maxit,
                       prev_init=st[t0 - 1] if t0 > 0 else None)
            ran.append(h.get_option("last_overlap"))
            tip[:, t0:t0 + n] = tp; status[:, t0:t0 + n] = ss; t0 += n
        last = st[T].double().cpu().numpy()
    torch.cuda.synchronize()
    return tip.cpu().numpy(), status.cpu().numpy(), last, ran, h.get_option("last_waves_per_rod")

bad = 0
for case in range(cases):
    N = int(rng.choice([33, 40, 57, 64, 90, 100, 101, 128, 200, 400])); W = int(rng.choice([2, 4])); B = int(rng.choice([1, 3, 16, 64]))
    T = int(rng.choice([1, 2, 5, 24, 60])); kind = str(rng.choice(["random", "jumps", "sine"])); ring = bool(rng.integers(0, 2))
    chunks = int(rng.choice([1, 1, 3])) if T >= 6 else 1; maxit = int(rng.choice([0, 0, 0, 2, 3])); mod = mods[int(rng.integers(0, len(mods)))]
    if chunks > 1: ring = False
    r = CosseratRod(use_fsolve=True); setup_robot(r, mod); r.N = N; r.compute_intermediate_terms(); h = r._native()
    h.set_option("waves_per_rod", W)
    ctl = controls(kind, B, T)
    dt = torch.float32 if rng.integers(0, 3) == 0 else torch.float64
    a = run(h, ctl, B, T, ring, chunks, 0, maxit, dt); b = run(h, ctl, B, T, ring, chunks, 1, maxit, dt)
    ok_rows = (a[1] == 0).all(axis=1) & (b[1] == 0).all(axis=1)   # rods that converged on every step in both runs
    scale = np.abs(a[0]).max() + 1e-30
    terr = np.abs(a[0][ok_rows] - b[0][ok_rows]).max() / scale if ok_rows.any() else 0.0
    serr = np.abs(a[2][ok_rows][..., :25] - b[2][ok_rows][..., :25]).max() / (np.abs(a[2][..., :25]).max() + 1e-30) if ok_rows.any() else 0.0
    same_status = (a[1] == b[1]).all()
    f32 = dt == torch.float32
    ok = terr < (2e-4 if f32 else 5e-7) and serr < (2e-3 if f32 else 5e-6) and same_status and np.isfinite(b[0][ok_rows]).all()
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} case {case}: {'f32' if f32 else 'f64'} mod={mod} N={N} W={W}->{b[4]} B={B} T={T} {kind} ring={int(ring)} chunks={chunks} maxit={maxit} overlap_ran={b[3]} "
          f"rods converged in both {int(ok_rows.sum())}/{B} unconverged steps {int((a[1] != 0).sum())}/{int((b[1] != 0).sum())} same status {bool(same_status)} tip err {terr:.2e} state err {serr:.2e}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
