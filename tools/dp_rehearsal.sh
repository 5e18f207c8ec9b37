#!/bin/bash
# GPU box (1 GPU): the data-parallel training path with 2 ranks sharing the GPU over gloo must reproduce the loss curve
# of one process that holds both trajectories.
set -e
export PYTHONPATH=knode-cosserat_amd
timeout -k 10 300 python3 knode-cosserat_amd/train_knode.py sine random 2 7 --fast --mod damping --epochs 31 --layers 32 --no-eval --save gpurun_out/dp1.pth | grep "Total loss" > gpurun_out/dp1.txt
KR_DIST_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 \
  knode-cosserat_amd/train_knode.py sine random 2 7 --fast --mod damping --epochs 31 --layers 32 --no-eval --save gpurun_out/dp2.pth 2>/dev/null | grep "Total loss" > gpurun_out/dp2.txt
paste gpurun_out/dp1.txt gpurun_out/dp2.txt
python3 - <<'PY'
import re
a=[float(re.search(r"Total loss: ([-0-9.e]+)",l).group(1)) for l in open("gpurun_out/dp1.txt")]
b=[float(re.search(r"Total loss: ([-0-9.e]+)",l).group(1)) for l in open("gpurun_out/dp2.txt")]
assert len(a)==len(b)==4, (a,b)
assert all(abs(x-y)<=1e-4*abs(x) for x,y in zip(a,b)), (a,b)
print("data-parallel loss curve == single-process loss curve")
PY
# Rehearsal of the driver's multi-GPU command on the one-GPU box: bench.py --gpus 4 with the gloo backend (collectives on CPU
# tensors), four ranks sharing the card (the pool's guard allows 6 processes on one GPU and counts the launcher too: six
# ranks were killed by it); the driver's own run is --gpus 8 with nccl = RCCL, one rank per GPU.  Same code path: rank-sharded
# rods, barrier + MAX timing, weak and strong scaling, the cfg4 data-parallel leg with one all-reduce per epoch and the
# single-rank repeat.  JSON -> gpurun_out/<tag>_rehearsal4.json
tag=${1:-r05}
KR_BENCH_BACKEND=gloo timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu > gpurun_out/${tag}_rehearsal4.json 2> gpurun_out/${tag}_rehearsal4.err
tail -c 1500 gpurun_out/${tag}_rehearsal4.json
