#!/usr/bin/env python3
"""GPU box: BASELINE cfg5 alone (B = 512, N = 400, fp64, sine tensions) - the SAME leg bench.py reports as extra.cfg5
(bench_legs.forward_leg: 30 untimed + 60 timed steps of one trajectory per rod, two kr_simulate_batch calls, predictor
handed over), run once, so that a rocprofv3 trace holds exactly two dispatches of the step kernel: the untimed and the
timed one (tools/promote_r04.py quotes the LAST).    python tools/cfg5_only.py [f64|f32]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import torch
import bench_legs as bl
dt = "f32" if (len(sys.argv) > 1 and sys.argv[1] == "f32") else "f64"
r = bl.forward_leg(torch, 0, 512, 400, 60, 30, dt, 1237, repeats=1)
print(f"cfg5 {dt}: {r['ms_per_step'] * 1e3:.1f} us/step wall, {r['kernel_ms_per_step'] * 1e3:.1f} us/step by HIP events, "
      f"{r['value'] / 1e6:.2f} M rod-steps/s, {r['kernel']}, unconverged {r['unconverged']}")
