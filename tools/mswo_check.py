#!/usr/bin/env python3
"""GPU box: the several-wavefront kernel with overlapped steps (option msw_overlap, kr_mswo_impl.hpp) against the same
kernel family without the overlap: states / tips / status of short runs, then the timing of BASELINE cfg2.
    python tools/mswo_check.py [quick]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench_legs as bl, krod_native as kn
dev = "cuda:0"
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"

def run(N, B, W, T, dt, overlap, seed=5, full=True, warm=0):
    r = bl.make_robot(N, 0); h = r._native()
    h.set_option("waves_per_rod", W); h.set_option("msw_overlap", overlap)
    ctl = torch.as_tensor(bl.sine_controls(B, warm + T, r.del_t, seed), device=dev).to(dt).contiguous()
    st = h.new_state(B, dt, n_slots=(warm + T + 1) if full else 3); h.init_straight(st[0])
    G = torch.zeros((B, 6), dtype=dt, device=dev)
    tip = torch.empty((B, warm + T, 3), dtype=dt, device=dev); status = torch.zeros((B, warm + T), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.simulate(ctl, st, G, ring=not full, tip=tip, status=status)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    return (st.double().cpu().numpy(), tip.double().cpu().numpy(), status.cpu().numpy(), G.double().cpu().numpy(), el,
            (h.get_option("last_sim_path"), h.get_option("last_waves_per_rod"), h.get_option("last_overlap")))

cases = [(40, 4, 2, 12, torch.float64), (100, 8, 4, 30, torch.float64)] if quick else \
        [(40, 8, 2, 40, torch.float64), (100, 32, 4, 60, torch.float64), (100, 64, 2, 60, torch.float64), (64, 32, 4, 50, torch.float32),
         (128, 16, 4, 40, torch.float64), (57, 16, 2, 40, torch.float64)]
for N, B, W, T, dt in cases:
    a = run(N, B, W, T, dt, 0); b = run(N, B, W, T, dt, 1)
    ds = np.abs(a[0][..., :25] - b[0][..., :25]).max() / np.abs(a[0][..., :25]).max()
    dtip = np.linalg.norm(a[1] - b[1]) / np.linalg.norm(a[1])
    print(f"N={N} B={B} W={W} T={T} {str(dt)[6:]}: paths {a[5]} / {b[5]}; unconverged {int((a[2] != 0).sum())} / {int((b[2] != 0).sum())}; "
          f"states max rel diff {ds:.2e}, tips rel L2 {dtip:.2e}, G diff {np.abs(a[3] - b[3]).max():.2e}", flush=True)
    a2 = run(N, B, W, T, dt, 0, full=False); b2 = run(N, B, W, T, dt, 1, full=False)
    print(f"   3-slot ring: tips rel L2 {np.linalg.norm(a2[1] - b2[1]) / np.linalg.norm(a2[1]):.2e}; against the full run {np.linalg.norm(b2[1] - b[1]) / np.linalg.norm(b[1]):.2e}; "
          f"final states rel diff {np.abs(a2[0][T % 3][..., :25] - b2[0][T % 3][..., :25]).max() / np.abs(a2[0][T % 3][..., :25]).max():.2e}", flush=True)
if not quick:
    for ov in (0, 1, 0, 1):
        r = bl.make_robot(100, 0); h = r._native(); h.set_option("msw_overlap", ov)
        leg = None
        import bench
        fn, args, kw = bench.extra_legs()["cfg2"]
        # (forward_leg builds its own robot: the option travels by the environment)
        os.environ["KR_MSW_OVERLAP"] = str(ov)
        leg = getattr(bl, fn)(torch, 0, *args, **kw)
        print(f"cfg2 msw_overlap={ov}: {leg['kernel_ms_per_step'] * 1e3:.2f} us/step by events, {leg['value'] / 1e6:.2f} M rod-steps/s, {leg['kernel']}, unconverged {leg['unconverged']}", flush=True)
