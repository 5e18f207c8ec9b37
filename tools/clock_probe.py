#!/usr/bin/env python3
"""Dev tool (GPU box, diagnostic library): does a SHORT persistent launch run at a lower shader clock?  Steady-state
trajectory (B=1024, N=100, fp64), K timed steps in one launch: wall time per step against the s_memtime ticks the
slowest / the average wavefront spent.   KR_LIB_PATH=knode-cosserat_amd/lib/dbg/libknode_rod.so python tools/clock_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench, krod_native as kn
from cosserat_ode import CosseratRod
from knode import setup_robot
B, N = 1024, 100
dev = "cuda:0"; dt = torch.float64
r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms()
h = r._native(); h.set_option("keep_predictor", 1)
TRACE_MAX = 1000
dbg = torch.zeros((B * 24 + 2 * TRACE_MAX,), dtype=torch.int64, device=dev)
for K in (20, 20, 100, 1000, 20):
    ctl = torch.as_tensor(bench.rank_controls(B, 1, 0, 60 + K, r.del_t), device=dev).contiguous()
    st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
    kn.check(h.lib.kr_debug_buffer(h._h, None))
    PRE = int(os.environ.get("KR_PRE", "60"))
    if os.environ.get("KR_PRE_PERSTEP"): h.set_option("persistent", 0)   # bench.py: untimed steps one launch per step
    if PRE != 60: ctl = torch.as_tensor(bench.rank_controls(B, 1, 0, PRE + K, r.del_t), device=dev).contiguous()
    h.simulate(ctl[:, :PRE].contiguous(), st, G, ring=True)
    h.set_option("persistent", 1)
    ck = ctl[:, PRE:].contiguous(); pi = st[(PRE + 2) % 3].clone(); st = st[[PRE % 3, (PRE + 1) % 3, (PRE + 2) % 3]].contiguous()
    dbg.zero_()
    dbg[(B - 1) * 24 + 15] = 0xC10C
    kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(dbg)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); h.simulate(ck, st, G, ring=True, prev_init=pi); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    tr = dbg[B * 24:B * 24 + 2 * K].cpu().numpy().reshape(K, 2).astype(np.float64)
    d = dbg[:B * 24].reshape(B, 24).cpu().numpy().astype(np.float64)
    tot, its = d[:, 0], d[:, 4]
    print(f"K={K:5d}: {ms*1e3/K:6.1f} us/step (event) | ticks/step mean {tot.mean()/K:8.0f} max {tot.max()/K:8.0f} | sweeps/step mean {its.mean()/K:.3f} max {its.max()/K:.3f} | "
          f"max ticks / event time = {tot.max()/(ms*1e3):7.1f} ticks/us")
    qn = dbg[:B * 24].reshape(B, 24)[:, 15].cpu().numpy().view(np.float64)
    print(f"      per step: sweep {d[:,1].mean()/K:.0f} algebra {d[:,2].mean()/K:.0f} (hand-over {d[:,8].mean()/K:.0f} chain {d[:,9].mean()/K:.0f} solve {d[:,10].mean()/K:.0f} rest {d[:,11].mean()/K:.0f}) "
          f"history+guess {d[:,3].mean()/K:.0f}; residual-test acceptances per step {qn.mean()/K:.3f}; slowest rod: sweep {d[:,1].max()/K:.0f} algebra {d[:,2].max()/K:.0f}")
    if K >= 2:
        dc, dr = np.diff(tr[:, 0]), np.diff(tr[:, 1])           # shader cycles and 10 ns units per step of rod 0
        mhz = dc / (dr * 0.01)
        pick = sorted(set([0, 1, 2, 3, 5, 8, 12, 18] + list(range(24, K - 1, max(1, (K - 1) // 12)))))
        print("      shader clock over the launch (MHz at step): " + " ".join(f"{i}:{mhz[i]:.0f}" for i in pick if i < K - 1), flush=True)
