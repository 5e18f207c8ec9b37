#!/bin/bash
# GPU box: everything a round's profiles/ directory is made of.  ONE script for all rounds (it replaces collect_profiles.sh,
# collect_r02/r03/r04.sh, prof_train.sh and pmc_train.sh, whose overlapping output names let one run overwrite another's
# raw files - VERDICT round 4, weak #1).  Every pass writes to its OWN directory gpurun_out/<tag>_<section>_<pass>; nothing
# is removed or reused across sections.  tools/profile_summary.py <tag> turns the raw output into profiles/<tag>_*.
#   usage: bash tools/collect.sh <tag> [section ...]      sections (default: all):
#     bench   bench line with the CPU baseline          nn      MLP-on simulate (cfg3 forward): stats + SQ / MFMA counters
#     stats   rocprofv3 kernel stats of bench.py        train   training epoch (cfg3, cfg4 shard): stats + SQ + HBM counters
#     hbm     FETCH_SIZE / WRITE_SIZE passes            ode     batched ODE kernel
#     sq      SQ issue counters                         cfg5    N = 400 leg: stats + HBM + SQ + fp64 op counters
#     ops     executed fp64 / MFMA op counters          cfg2    B = 256 leg: stats + SQ + fp64 op counters
#     configs tools/config_report.py over all five BASELINE configurations
tag=${1:?usage: collect.sh <tag> [section ...]}
shift
sections=${*:-bench stats hbm sq ops nn train ode cfg5 cfg2 configs}
out=gpurun_out
export TMPDIR=/tmp
mkdir -p $out
want() { [[ " $sections " == *" $1 "* ]]; }
# one rocprofv3 pass: prof <dir suffix> <seconds> <rocprof args ...> -- <program ...>; stdout / stderr of the program are kept
prof() {
  local name=$1 secs=$2; shift 2
  local d=$out/${tag}_$name
  if [ -e "$d" ]; then echo "refusing to overwrite $d (another run of this tag wrote it)"; return 1; fi
  timeout -k 10 $secs rocprofv3 --kernel-trace "$@" > $d.out 2> $d.err || echo "pass $name failed (rc $?)"
}
SQ_ISSUE="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"
SQ_MFMA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS"
SQ_MEM="SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
# executed arithmetic (names checked against `rocprofv3 -L` on the box: gpurun_out/<tag>_counters.txt)
OPS_F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
OPS_MFMA="SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"
OPS_F32="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES"
BENCH="python3 bench.py --no-cpu --no-extra"
rocprofv3 -L > $out/${tag}_counters.txt 2>&1 || true

# one leg of bench.py's `extra` under the profiler: leg_passes <leg> <pass ...>   (pass: stats fetch write sq sqm ops opsm vops64 vops32)
leg_passes() {
  local leg=$1; shift
  for p in "$@"; do
    case $p in
      stats) prof ${leg}_stats 300 --stats --output-format csv -d $out/${tag}_${leg}_stats -- python3 tools/leg_only.py $leg
             grep -h "^leg " $out/${tag}_${leg}_stats.out ;;
      fetch) prof ${leg}_fetch 300 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_${leg}_fetch -- python3 tools/leg_only.py $leg ;;
      write) prof ${leg}_write 300 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_${leg}_write -- python3 tools/leg_only.py $leg ;;
      sq)    prof ${leg}_sq 300 --pmc $SQ_ISSUE --output-format csv -d $out/${tag}_${leg}_sq -- python3 tools/leg_only.py $leg ;;
      sqm)   prof ${leg}_sq 300 --pmc $SQ_MFMA --output-format csv -d $out/${tag}_${leg}_sq -- python3 tools/leg_only.py $leg ;;
      ops)   prof ${leg}_ops 300 --pmc $OPS_F64 --output-format csv -d $out/${tag}_${leg}_ops -- python3 tools/leg_only.py $leg ;;
      opsm)  prof ${leg}_ops 300 --pmc $OPS_MFMA --output-format csv -d $out/${tag}_${leg}_ops -- python3 tools/leg_only.py $leg ;;
      vops64) prof ${leg}_vops 300 --pmc $OPS_F64 --output-format csv -d $out/${tag}_${leg}_vops -- python3 tools/leg_only.py $leg ;;
      vops32) prof ${leg}_vops 300 --pmc $OPS_F32 --output-format csv -d $out/${tag}_${leg}_vops -- python3 tools/leg_only.py $leg ;;
    esac
  done
}
if want bench; then
  echo "[bench] bench line (with CPU baseline)"
  timeout -k 10 500 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { echo "bench failed"; exit 1; }
  tail -c 400 $out/${tag}_bench.json; echo
  # the driver's own command line (20-step chunks): the fixed cost of a call shows here
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $out/${tag}_bench_steps20.json 2> $out/${tag}_bench_steps20.err || true
fi
if want stats; then
  echo "[stats] rocprofv3 kernel stats of the same command (--no-cpu --no-extra)"
  prof bench_stats 300 --stats --output-format csv -d $out/${tag}_bench_stats -- $BENCH
fi
if want hbm; then
  echo "[hbm] HBM counters, separate passes"
  prof bench_fetch 300 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_bench_fetch -- $BENCH
  prof bench_write 300 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_bench_write -- $BENCH
  # the full-trajectory mode of the same kernel: (25 N + 4) s algorithmic bytes per rod-step
  leg_passes headline_full_trajectory stats fetch write
fi
if want sq; then
  echo "[sq] SQ issue counters"
  prof bench_sq 300 --pmc $SQ_ISSUE --output-format csv -d $out/${tag}_bench_sq -- $BENCH
fi
if want ops; then
  echo "[ops] executed fp64 operation counters of the headline kernel"
  prof bench_ops 300 --pmc $OPS_F64 --output-format csv -d $out/${tag}_bench_ops -- $BENCH
fi
if want nn; then
  echo "[nn] MLP-on simulate (cfg3 forward), fp64 and fp32: stats, SQ + matrix-pipe counters, executed matrix operations"
  leg_passes cfg3_nn_f64 stats sqm opsm vops64
  leg_passes cfg3_nn_f32 stats sqm opsm vops32
fi
if want train; then
  echo "[train] training epoch kernels (cfg3: 28-64-64-25, Q = 193 536; cfg4 shard: 28-512-25, Q = 59 392)"
  for cfg in cfg3 cfg4; do
    prof train_${cfg}_stats 300 --stats --output-format csv -d $out/${tag}_train_${cfg}_stats -- python3 tools/train_only.py $cfg
    prof train_${cfg}_sq1 300 --pmc $SQ_MFMA --output-format csv -d $out/${tag}_train_${cfg}_sq1 -- python3 tools/train_only.py $cfg
    prof train_${cfg}_sq2 300 --pmc $SQ_MEM --output-format csv -d $out/${tag}_train_${cfg}_sq2 -- python3 tools/train_only.py $cfg
    prof train_${cfg}_fetch 300 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_train_${cfg}_fetch -- python3 tools/train_only.py $cfg
    prof train_${cfg}_write 300 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_train_${cfg}_write -- python3 tools/train_only.py $cfg
  done
  prof train_literal_stats 300 --stats --output-format csv -d $out/${tag}_train_literal_stats -- python3 tools/leg_only.py cfg3_mlp_literal
fi
if want ode; then
  echo "[ode] batched ODE kernel"
  timeout -k 10 200 python3 tools/aux_bench.py ode > $out/${tag}_ode.txt 2>&1; grep ode_batch $out/${tag}_ode.txt
  prof ode_stats 300 --stats --output-format csv -d $out/${tag}_ode_stats -- python3 tools/aux_bench.py ode
fi
if want cfg5; then
  echo "[cfg5] N = 400, B = 512 (same leg as bench.py extra.cfg5)"
  leg_passes cfg5 stats fetch write sq ops
fi
if want cfg2; then
  echo "[cfg2] N = 100, B = 256 (same leg as bench.py extra.cfg2)"
  leg_passes cfg2 stats sq ops
fi
if want configs; then
  echo "[configs] all five BASELINE configurations"
  timeout -k 10 600 python3 tools/config_report.py > $out/${tag}_configs.txt 2>&1; grep -v amdgpu $out/${tag}_configs.txt
fi
echo collected
