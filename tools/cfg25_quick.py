#!/usr/bin/env python3
"""GPU box: cfg2 (B = 256, N = 100) and cfg5 (B = 512, N = 400) legs of bench.py on their own.  python tools/cfg25_quick.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import torch
import bench_legs as bl
for name, a in (("cfg2", (256, 100, 200, 60, "f64", 1234)), ("cfg2 B=512", (512, 100, 200, 60, "f64", 1234)), ("cfg5", (512, 400, 60, 30, "f64", 1237)), ("cfg5 B=256", (256, 400, 60, 30, "f64", 1237))):
    r = bl.forward_leg(torch, 0, *a)
    print(f"{name}: {r['ms_per_step']*1e3:.2f} us/step, {r['value']/1e6:.3f} M rod-steps/s, {r['kernel']}, unconverged {r['unconverged']}", flush=True)
