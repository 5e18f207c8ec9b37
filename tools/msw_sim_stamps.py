#!/usr/bin/env python3
"""Dev tool (GPU box): sweep statistics of the PERSISTENT several-wavefront kernel (msw_sim_kernel, MLP off) on a bench leg's
workload: `warm` untimed + T timed steps, predictor handed over - what extra.cfg2 / extra.cfg5 time.  Needs the diagnostic
library (make -C knode-cosserat_amd/csrc dbg [DBGFLAGS="-DKR_MS_STAMPS -DKR_MSW_TAPS=5"]):
    KR_LIB_PATH=knode-cosserat_amd/lib/dbg/libknode_rod.so python tools/msw_sim_stamps.py N B [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench_legs as bl, krod_native as kn
N, B = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else (1237 if N == 400 else 1234)
warm, T = (30, 60) if N == 400 else (60, 200)
dev = "cuda:0"; dt = torch.float64
r = bl.make_robot(N, 0); h = r._native()
ctl = torch.as_tensor(bl.sine_controls(B, warm + T, r.del_t, seed), device=dev).to(dt).contiguous()
dbg = torch.zeros((B, 24), dtype=torch.int64, device=dev)
h.set_option("keep_predictor", 1)
for rep in range(3):
    st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
    kn.check(h.lib.kr_debug_buffer(h._h, None))
    h.simulate(ctl[:, :warm].contiguous(), st, G, ring=True)
    newest, older = st[warm % 3].clone(), st[(warm - 1) % 3].clone()
    st[0].copy_(newest)
    dbg.zero_(); kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(dbg)))
    status = torch.zeros((B, T), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.simulate(ctl[:, warm:].contiguous(), st, G, ring=True, status=status, prev_init=older)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
d = dbg.cpu().numpy().astype(np.float64)
tot, sw, al, its = d[:, 0], d[:, 1], d[:, 2], d[:, 4]
print(f"N={N} B={B} W={h.get_option('last_waves_per_rod')} path {h.get_option('last_sim_path')}: wall {el/T*1e6:.1f} us/step (diagnostic build); unconverged {int((status != 0).sum())}")
print(f"  ticks per step: total mean {tot.mean()/T:.0f} max {tot.max()/T:.0f} | sweep {sw.mean()/T:.0f} algebra {al.mean()/T:.0f} | sweeps per step mean {its.mean()/T:.3f}, slowest rod {its.max()/T:.3f}")
its_sum = max(its.sum(), 1.0)
print(f"  steps with <= 2 / 3 / >= 4 sweeps: {d[:,5].sum()/(B*T):.3f} / {d[:,6].sum()/(B*T):.3f} / {d[:,7].sum()/(B*T):.3f}; per sweep {sw.sum()/its_sum:.0f} ticks, algebra per sweep {al.sum()/its_sum:.0f}")
print(f"  condensation per step: local chains {d[:,8].mean()/T:.0f}, barrier {d[:,9].mean()/T:.0f}, boundary chain {d[:,10].mean()/T:.0f}, solve + back-substitution + p rows + norms + decisions {d[:,11].mean()/T:.0f}")
print(f"  step loop per step: history build {d[:,12].mean()/T:.0f}, start values {d[:,13].mean()/T:.0f}, Newton (sweeps + algebra + residual tests) {d[:,14].mean()/T:.0f}, predictor update {d[:,15].mean()/T:.0f}")
print(f"  inside the last figure of the condensation: 6 x 6 solve {d[:,16].mean()/T:.0f}, back-substitution + barrier {d[:,17].mean()/T:.0f}, p rows + barrier {d[:,18].mean()/T:.0f}, norms + reduction over the rod {d[:,19].mean()/T:.0f}")
