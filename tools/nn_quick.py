#!/usr/bin/env python3
"""GPU box: the two NN-on legs of bench.py (cfg3: B = 1024, N = 100, 28 -> 64 -> 64 -> 25, T = 64 from the straight rod) and the
same at B = 512 / 256, without the rest of the bench.    python tools/nn_quick.py [B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import torch
import bench_legs as bl
mlp = bl.mlp_weights([28, 64, 64, 25], 7)
for B in [int(a) for a in sys.argv[1:]] or [1024]:
    for dt in ("f64", "f32"):
        r = bl.forward_leg(torch, 0, B, 100, 64, 0, dt, 1235, mlp=mlp, repeats=2)
        print(f"B={B} {dt}: {r['ms_per_step']:.4f} ms/step, {r['value']/1e6:.3f} M rod-steps/s, {r['kernel']}, W={r['waves_per_rod']}, unconverged {r['unconverged']}", flush=True)
