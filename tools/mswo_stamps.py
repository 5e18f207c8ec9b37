#!/usr/bin/env python3
"""Dev tool (GPU box): counters of the overlapped several-wavefront kernel on a leg's workload (diagnostic build of
kr_mswo_f64.hip with -DKR_MS_STAMPS).   KR_LIB_PATH=.../dbg/libknode_rod_mswo.so python tools/mswo_stamps.py N B"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench_legs as bl, krod_native as kn
N, B = int(sys.argv[1]), int(sys.argv[2])
warm, T = 60, 200
dev = "cuda:0"; dt = torch.float64
r = bl.make_robot(N, 0); h = r._native(); h.set_option("msw_overlap", 1); h.set_option("keep_predictor", 1)
ctl = torch.as_tensor(bl.sine_controls(B, warm + T, r.del_t, 1234), device=dev).to(dt).contiguous()
dbg = torch.zeros((B, 24), dtype=torch.int64, device=dev)
st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
h.simulate(ctl[:, :warm].contiguous(), st, G, ring=True)
newest, older = st[warm % 3].clone(), st[(warm - 1) % 3].clone(); st[0].copy_(newest)
kn.check(h.lib.kr_debug_buffer(h._h, kn._ptr(dbg)))
status = torch.zeros((B, T), dtype=torch.int32, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
h.simulate(ctl[:, warm:].contiguous(), st, G, ring=True, status=status, prev_init=older)
torch.cuda.synchronize(); el = time.perf_counter() - t0
d = dbg.cpu().numpy().astype(np.float64)
print(f"N={N} B={B} W={h.get_option('last_waves_per_rod')} overlap={h.get_option('last_overlap')}: wall {el/T*1e6:.1f} us/step; unconverged {int((status != 0).sum())}")
print(f"  per step: ticks {d[:,0].mean()/T:.0f} (sweep {d[:,1].mean()/T:.0f}, verdict {d[:,3].mean()/T:.0f}, condense {d[:,2].mean()/T:.0f}, hand-over+predictor {d[:,10].mean()/T:.0f}); "
      f"sweeps {d[:,4].mean()/T:.3f} (merged {d[:,5].mean()/T:.3f}); accepted {d[:,6].mean()/T:.3f} rejected {d[:,7].mean()/T:.3f} plain steps {d[:,8].mean()/T:.3f} accepted-after-rejection {d[:,9].mean()/T:.3f}")
rej = d[:, 7]; tot = d[:, 0]
print(f"  slowest rod: {tot.max()/T:.0f} ticks per step ({tot.max()/tot.mean():.2f} x the mean); rejections per rod: mean {rej.mean():.1f} max {rej.max():.0f} of {T} steps; "
      f"rods without any: {(rej == 0).mean():.2f}; chord checks per step {d[:,11].mean()/T:.3f} (max {d[:,11].max()/T:.3f}); ticks per step of those {tot[rej == 0].mean()/T if (rej == 0).any() else float('nan'):.0f}; tick rate {tot.max()/el/1e6:.0f} MHz")
