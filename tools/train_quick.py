#!/usr/bin/env python3
"""GPU box: the two training legs of bench.py (cfg3 epoch, cfg4 shard epoch) on their own.  python tools/train_quick.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import torch
import bench_legs as bl
for name, a in (("cfg3", (1024, 64, 100, [22, 67, 99], [64, 64])), ("cfg4", (512, 30, 10, [3, 5, 7, 9], [512]))):
    r = bl.train_leg(torch, 0, *a, epochs=40)
    print(f"{name}: {r['us_per_epoch']:.1f} us/epoch (min {r['us_per_epoch_min']:.1f}, wall {r['wall_us_per_epoch']:.1f}), {r['roofline']['achieved']:.1f} TF, "
          f"loss {r['loss_first']:.5f} -> {r['loss_last']:.5f}", flush=True)
