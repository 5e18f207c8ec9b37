#!/usr/bin/env python3
"""GPU box: one table over the five BASELINE.json configurations - throughput of the HIP path and
its accuracy against the golden vectors the reference produced.  Output is committed as profiles/<tag>_configs.txt.
    python tools/config_report.py > gpurun_out/configs.txt
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "knode-cosserat_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import torch.nn as nn
import cosserat_oracle as orc
from cosserat_ode import CosseratRod
from cosserat_ode_torch import CosseratRodTorch
from knode import setup_robot, simulate, simulate_batch
from krod_train import KnodeTrainer
dev = "cuda:0"
G = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))

def robot(N, mod=None):
    r = CosseratRod(use_fsolve=True); setup_robot(r, mod); r.N = N; r.compute_intermediate_terms(); return r

def timed_sim(r, B, T, dtype, seed, warm=60):
    h = r._native()
    ctl = torch.as_tensor(orc.batch_sine_controls(B, warm + T, r.del_t, seed), device=dev).to(dtype).contiguous()
    best = 1e9
    for _ in range(3):
        st = h.new_state(B, dtype, n_slots=3); h.init_straight(st[0])
        Gs = torch.zeros((B, 6), dtype=dtype, device=dev)
        status = torch.zeros((B, T), dtype=torch.int32, device=dev)
        h.simulate(ctl[:, :warm].contiguous(), st, Gs, ring=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.simulate(ctl[:, warm:].contiguous(), st, Gs, ring=True, status=status, prev_init=st[2])
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    path = h.get_option("last_sim_path")
    w = h.get_option("last_waves_per_rod")
    return best / T, int((status != 0).sum()), (f"{path}, {w} wavefronts per rod" if (path == 1 or w > 1) else path)

print("cfg1  single rod, N=20, 200 steps, constant tensions (reference fixture sim_cfg1)")
g = G("sim_cfg1"); r = robot(20)
t0 = time.perf_counter(); tr = simulate(r, g["ctl"]); el = time.perf_counter() - t0
ws = []
for _ in range(5):
    t0 = time.perf_counter(); tr = simulate(r, g["ctl"]); ws.append(time.perf_counter() - t0)
print(f"      tip rel L2 vs reference {rel(tr[:, :3, -1], g['tip']):.2e}; knode.simulate wall, 200 steps incl. host<->device copies: "
      f"first call {el*1e3:.1f} ms (library load, handle, allocations), then {np.median(ws)*1e3:.2f} ms = {np.median(ws)/200*1e3:.4f} ms/step")

print("cfg2  B=256, N=100, forward only")
g = G("sim_n100"); r = robot(100)
out = simulate_batch(r, g["batch_ctl"]); ref = g["batch_tip"]
e64 = max(rel(out["traj"][b, :ref.shape[1], :3, -1], ref[b]) for b in range(ref.shape[0]))
out32 = simulate_batch(r, g["batch_ctl"], dtype="f32")
e32 = max(rel(out32["traj"][b, :ref.shape[1], :3, -1], ref[b]) for b in range(ref.shape[0]))
for dt in (torch.float64, torch.float32):
    s, bad, path = timed_sim(r, 256, 200, dt, 1234)
    print(f"      {str(dt):14s} {s*1e6:7.1f} us/step -> {256/s/1e6:6.2f} M rod-steps/s (path {path}, unconverged {bad}); tip rel L2 vs reference (6 rods x 24 steps) {e64 if dt==torch.float64 else e32:.2e}")

print("cfg2' B=1024, N=100 (the bench.py workload)")
for dt in (torch.float64, torch.float32):
    s, bad, path = timed_sim(r, 1024, 400, dt, 1235)
    print(f"      {str(dt):14s} {s*1e6:7.1f} us/step -> {1024/s/1e6:6.2f} M rod-steps/s (path {path}, unconverged {bad})")

print("cfg3  B=1024, N=100, KNODE MLP 28->64->64->25: one-step-ahead forward+backward over Q=B*T*K rows; forward sim with the MLP on")
# (the training figures come from the legs bench.py reports - one methodology, one number: bench_legs.train_leg /
#  mlp_literal_leg, median of HIP-event times after a clock ramp)
sys.path.insert(0, ROOT)
import bench_legs as bl
def train_epoch(M, T, N, kp, layers):
    leg = bl.train_leg(torch, 0, M, T, N, kp, layers)
    return leg["us_per_epoch"] * 1e-6, leg["rows"], leg["roofline"]["achieved"] * 1e12 * leg["us_per_epoch"] * 1e-6, leg
t, Q, fl, leg = train_epoch(1024, 64, 100, [22, 67, 99], [64, 64])
print(f"      training epoch (fwd + loss + bwd + Adam + clamp), Q={Q} rows: {t*1e6:7.1f} us (median of {leg['timed_epochs']} epochs after {leg['ramp_epochs']}; {leg['wall_us_per_epoch']:.1f} us wall per epoch queued by kr_train_epochs) "
      f"-> {1024*63/t/1e6:6.1f} M trajectory-steps/s, {fl/t/1e12:5.1f} TFLOP/s fp32 useful")
lit = bl.mlp_literal_leg(torch, 0, 193536)
print(f"      bare MLP 18->64->64->6 (BASELINE-literal), forward + backward over Q=193536 rows: {lit['us_per_fwd_bwd']:7.1f} us, {lit['roofline']['achieved']:5.1f} TFLOP/s fp32 useful")
rr = robot(100); mlp = orc.make_mlp([28, 64, 64, 25], "elu", seed=7)
model, params = [], []
for W, b, a in zip(mlp.weights, mlp.biases, mlp.acts):
    model.append("Linear"); params += [W, b]
    if a != orc.ACT_NONE: model.append("ELU(alpha=1.0)")
rr.nn_model, rr.param_ls, rr.nn_path = model, params, "x"
h = rr._native()
for dt in (torch.float64, torch.float32):
    # one call (the start-value predictor lives for the length of a kr_simulate_batch call); 60 steps
    TS = 60
    ctl = torch.as_tensor(orc.batch_sine_controls(1024, TS, rr.del_t, 1235), device=dev).to(dt).contiguous()
    st = h.new_state(1024, dt, n_slots=3); h.init_straight(st[0]); Gs = torch.zeros((1024, 6), dtype=dt, device=dev)
    status = torch.zeros((1024, TS), dtype=torch.int32, device=dev)
    h.simulate(ctl, st, Gs, ring=True, use_nn=True, status=status)   # (first call of this instantiation: code load)
    st = h.new_state(1024, dt, n_slots=3); h.init_straight(st[0]); Gs.zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    h.simulate(ctl, st, Gs, ring=True, use_nn=True, status=status)
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / TS
    print(f"      forward sim, MLP inside every sweep, {str(dt):14s}: {el*1e3:6.3f} ms/step -> {1024/el/1e3:7.1f} k rod-steps/s (unconverged {int((status!=0).sum())})")
for Bs in (512, 256):  # smaller batches: several wavefronts per rod (kr_mswn_impl.hpp)
    for dt in (torch.float64, torch.float32):
        TS = 60
        ctl = torch.as_tensor(orc.batch_sine_controls(Bs, TS, rr.del_t, 1235), device=dev).to(dt).contiguous()
        st = h.new_state(Bs, dt, n_slots=3); h.init_straight(st[0]); Gs = torch.zeros((Bs, 6), dtype=dt, device=dev)
        status = torch.zeros((Bs, TS), dtype=torch.int32, device=dev)
        h.simulate(ctl, st, Gs, ring=True, use_nn=True, status=status)   # (first call of this instantiation)
        st = h.new_state(Bs, dt, n_slots=3); h.init_straight(st[0]); Gs.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h.simulate(ctl, st, Gs, ring=True, use_nn=True, status=status)
        torch.cuda.synchronize(); el = (time.perf_counter() - t0) / TS
        print(f"      ... the same at B={Bs:4d}, {str(dt):14s}: {el*1e3:6.3f} ms/step -> {Bs/el/1e3:7.1f} k rod-steps/s "
              f"({h.get_option('last_waves_per_rod')} wavefronts per rod, unconverged {int((status!=0).sum())})")

print("cfg4  training loop shard: 512 trajectories per GPU (4096 over 8), train_len 30, 28->512->25")
for N, kp in ((10, [3, 5, 7, 9]), (100, [33, 55, 77, 99])):
    t, Q, fl, _ = train_epoch(512, 30, N, kp, [512])
    print(f"      N={N:3d}: epoch {t*1e6:7.1f} us (Q={Q} rows) -> {512*29/t/1e6:6.1f} M trajectory-steps/s per GPU, {fl/t/1e12:5.1f} TFLOP/s fp32 useful; + one all-reduce of {28*512+512+512*25+25+1} floats per epoch")

print("cfg5  B=512, N=400, sine tensions; tolerance sweep on the reference fixture sim_n400 (single rod, 12 steps)")
g = G("sim_n400"); r = robot(400)
for dname in ("f64", "f32"):  # the leg bench.py reports as extra.cfg5 (one methodology, one number)
    leg = bl.forward_leg(torch, 0, 512, 400, 60, 30, dname, 1237)
    print(f"      {dname}  {leg['kernel_ms_per_step']*1e3:7.1f} us/step (HIP events; {leg['ms_per_step']*1e3:.1f} wall) -> {leg['value']/1e6:6.2f} M rod-steps/s "
          f"({leg['kernel']}, unconverged {leg['unconverged']})")
h = r._native()
for dname, dt, tols in (("f64", torch.float64, (1e-6, 1e-8, 1e-10, 1e-12)), ("f32", torch.float32, (1e-3, 1e-4, 1e-5, 1e-6))):
    line = []
    for tol in tols:
        c = torch.as_tensor(g["ctl"][None], device=dev).to(dt).contiguous()
        T = c.shape[1]
        st = h.new_state(1, dt, n_slots=T + 1); h.init_straight(st[0]); Gs = torch.zeros((1, 6), dtype=dt, device=dev)
        tip = torch.empty((1, T, 3), dtype=dt, device=dev); status = torch.zeros((1, T), dtype=torch.int32, device=dev)
        h.simulate(c, st, Gs, tip=tip, status=status, tol=tol); torch.cuda.synchronize()
        # the reference drops its last solve and lists the initial tip first (knode.py:96-102)
        got = np.concatenate([st[0][0, -1, 12:15].cpu().numpy()[None], tip[0].cpu().numpy()])[:T]
        line.append(f"tol {tol:.0e}: {rel(got.astype(np.float64), g['tip']):.1e}{'' if int((status!=0).sum())==0 else ' (!)'}")
    print(f"      {dname} tip rel L2 vs reference: " + "   ".join(line))
