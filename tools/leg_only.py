#!/usr/bin/env python3
"""GPU box: ONE leg of bench.py's `extra` object, alone, for profiling - the SAME function with the SAME arguments as
bench.run_extras uses (bench.EXTRA_LEGS), so that what rocprofv3 sees is what the driver line reports.  Every leg runs its
`repeats` timed calls; tools/profile_summary.py picks the timed dispatches by kernel instantiation AND duration.
    python tools/leg_only.py <cfg2|cfg3_nn_f64|cfg3_nn_f32|cfg5|cfg5_f32|headline_full> [repeats]"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd"))
import torch
import bench
import bench_legs as bl

name = sys.argv[1]
alias = {"headline_full": "headline_full_trajectory"}
name = alias.get(name, name)
legs = bench.extra_legs()
if name not in legs:
    raise SystemExit(f"unknown leg {name}; have {sorted(legs)}")
fn, args, kw = legs[name]
if len(sys.argv) > 2:
    kw = dict(kw, repeats=int(sys.argv[2]))
r = getattr(bl, fn)(torch, 0, *args, **kw)
print(f"leg {name}: {r.get('ms_per_step', 0) * 1e3:.1f} us/step wall, {r.get('kernel_ms_per_step', 0) * 1e3:.1f} us/step by HIP events "
      f"(best of {r.get('repeats')}), {r.get('kernel')}, unconverged {r.get('unconverged')}")
print("legjson " + json.dumps({k: v for k, v in r.items() if k != "roofline"}))
