#!/usr/bin/env python3
"""Dev tool (GPU box): the shader clock around a short persistent launch (B = 1024, N = 100, fp64, steady-state trajectory).
One script, three modes (formerly clock_probe.py / clock_probe2.py / clock_probe3.py):
    trace  K timed steps in one launch: wall time per step against the s_memtime ticks of the slowest / the average
           wavefront and the clock over the launch - needs the diagnostic library:
           KR_LIB_PATH=knode-cosserat_amd/lib/dbg/libknode_rod.so python tools/clock_probe.py trace
    ramp   how the KIND of load that precedes a 20-step launch sets its speed (none / fp32 / fp64 RK4 / fp64 Euler, 0.25 s)
    gap    what should sit between a heavy ramp and the timed 20-step launch (pause / light launches of varying length)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import numpy as np, torch
import bench, krod_native as kn
from cosserat_ode import CosseratRod
from knode import setup_robot
mode = sys.argv[1] if len(sys.argv) > 1 else "trace"
dev = "cuda:0"; dt = torch.float64


def trace():
    B, N = 1024, 100
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


def ramp_kind():
    B, N, K = 1024, 100, 20
    def robot():
        r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms(); return r
    r = robot(); h = r._native(); h.set_option("keep_predictor", 1)
    r2 = robot(); h2 = r2._native()
    ctl = torch.as_tensor(bench.rank_controls(B, 1, 0, 60 + K, r.del_t), device=dev).contiguous()
    def ramp(kind, seconds):
        if kind == "none": return
        d = torch.float32 if kind == "f32" else torch.float64
        scheme = kn.KR_RK4 if kind == "f64rk4" else kn.KR_EULER
        c = ctl[:, :60].to(d).repeat(1, 4, 1).contiguous()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            s2 = h2.new_state(B, d, n_slots=3); h2.init_straight(s2[0]); g2 = torch.zeros((B, 6), dtype=d, device=dev)
            h2.simulate(c, s2, g2, ring=True, scheme=scheme)
            torch.cuda.synchronize()
    for kind in ("none", "f32", "f64rk4", "f64", "none", "f64rk4"):
        res = []
        for rep in range(3):
            st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
            h.simulate(ctl[:, :60].contiguous(), st, G, ring=True)
            ck = ctl[:, 60:].contiguous(); pi = st[2].clone()
            ramp(kind, 0.25)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(); h.simulate(ck, st, G, ring=True, prev_init=pi); e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / K)
        print(f"ramp {kind:7s}: us/step of the 20-step launch: " + " ".join(f"{x:.1f}" for x in res), flush=True)


def gap():
    B, N, K = 1024, 100, 20
    def robot():
        r = CosseratRod(use_fsolve=True); setup_robot(r); r.N = N; r.compute_intermediate_terms(); return r
    r = robot(); h = r._native(); h.set_option("keep_predictor", 1)
    r2 = robot(); h2 = r2._native()
    ctl = torch.as_tensor(bench.rank_controls(B, 1, 0, 60 + K, r.del_t), device=dev).contiguous()
    c2 = ctl[:, :60].repeat(1, 2, 1)[:, :100].contiguous()
    s2 = h2.new_state(B, dt, n_slots=3); g2 = torch.zeros((B, 6), dtype=dt, device=dev)
    def heavy(seconds, sync_each):
        n = max(1, int(seconds / 0.012))
        for _ in range(n):
            h2.init_straight(s2[0]); g2.zero_()
            h2.simulate(c2, s2, g2, ring=True, scheme=kn.KR_RK4)
            if sync_each: torch.cuda.synchronize()
        torch.cuda.synchronize()
    def light(n):  # n short launches of the same solver (Euler, 5 steps) with host gaps
        for _ in range(n):
            h2.init_straight(s2[0]); g2.zero_()
            h2.simulate(c2[:, :5].contiguous(), s2, g2, ring=True)
            torch.cuda.synchronize()
    cases = [("heavy 1.0 s async", lambda: heavy(1.0, False)), ("heavy 1.0 s, sync each", lambda: heavy(1.0, True)),
             ("heavy 1.0 s + sleep 2 ms", lambda: (heavy(1.0, False), time.sleep(0.002))),
             ("heavy 1.0 s + sleep 20 ms", lambda: (heavy(1.0, False), time.sleep(0.02))),
             ("heavy 1.0 s + sleep 200 ms", lambda: (heavy(1.0, False), time.sleep(0.2))),
             ("heavy 1.0 s + 20 light launches", lambda: (heavy(1.0, False), light(20))),
             ("heavy 0.1 s async", lambda: heavy(0.1, False)), ("nothing", lambda: None), ("heavy 3 s async", lambda: heavy(3.0, False))]
    for name, fn in cases:
        res = []
        for rep in range(3):
            st = h.new_state(B, dt, n_slots=3); h.init_straight(st[0]); G = torch.zeros((B, 6), dtype=dt, device=dev)
            h.simulate(ctl[:, :60].contiguous(), st, G, ring=True)
            ck = ctl[:, 60:].contiguous(); pi = st[2].clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            fn()
            torch.cuda.synchronize()
            e0.record(); h.simulate(ck, st, G, ring=True, prev_init=pi); e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / K)
        print(f"{name:34s}: us/step of the 20-step launch: " + " ".join(f"{x:.1f}" for x in res), flush=True)


{"trace": trace, "ramp": ramp_kind, "gap": gap}[mode]()
