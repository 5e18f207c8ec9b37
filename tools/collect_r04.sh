#!/bin/bash
# GPU box: everything the round's profiles/ directory is made of.  Raw output -> gpurun_out/<tag>_*;
# tools/summarise_profiles.py <tag> turns the bench part into profiles/<tag>_{bench.json,bench_kernel_stats.csv,pmc_hbm.json,
# pmc_sq.json}; the rest is copied by tools/promote_r04.py.
#   usage: bash tools/collect_r04.sh r04a
tag=${1:-r04}
out=gpurun_out
export TMPDIR=/tmp
mkdir -p $out
set -e
echo "[1] bench line (with CPU baseline)"
timeout -k 10 500 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 600 $out/${tag}_bench.json; echo
echo "[2] rocprofv3 kernel stats of the same command (--no-cpu)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --no-cpu --no-extra > $out/${tag}_stats.json 2> $out/${tag}_stats.err
echo "[3] HBM counters, separate passes"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 bench.py --no-cpu --no-extra > /dev/null 2> $out/${tag}_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 bench.py --no-cpu --no-extra > /dev/null 2> $out/${tag}_write.err
echo "[4] SQ counters"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $out/${tag}_sq -- python3 bench.py --no-cpu --no-extra > /dev/null 2> $out/${tag}_sq.err || echo "SQ pass failed (non-fatal)"
echo "[5] MLP-on simulate (cfg3 forward): kernel stats + matrix-pipe counters, fp64 and fp32"
for p in f64 f32; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_nn_${p}_stats -- python3 tools/simnn_only.py $p 40 > $out/${tag}_nn_${p}.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $out/${tag}_nn_${p}_sq -- python3 tools/simnn_only.py $p 40 > /dev/null 2>&1 || echo "nn SQ pass failed (non-fatal)"
done
echo "[6] training epoch kernels"
bash tools/prof_train.sh ${tag} > $out/${tag}_train.txt 2>&1; cat $out/${tag}_train.txt
for cfg in cfg3 cfg4; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/${tag}_train_${cfg}_sq -- python3 tools/train_only.py $cfg > /dev/null 2>&1 || echo "train SQ pass failed (non-fatal)"
done
echo "[7] batched ODE kernel"
timeout -k 10 200 python3 tools/aux_bench.py ode > $out/${tag}_ode.txt 2>&1; grep ode_batch $out/${tag}_ode.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_ode_stats -- python3 tools/aux_bench.py ode > /dev/null 2>&1 || true
echo "[9] cfg5 (N = 400, B = 512): the several-wavefront step kernel"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_cfg5_stats -- python3 tools/cfg5_only.py > $out/${tag}_cfg5.log 2>&1 || true
grep cfg5 $out/${tag}_cfg5.log
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_cfg5_fetch -- python3 tools/cfg5_only.py > /dev/null 2>&1 || true
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_cfg5_write -- python3 tools/cfg5_only.py > /dev/null 2>&1 || true
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $out/${tag}_cfg5_sq -- python3 tools/cfg5_only.py > /dev/null 2>&1 || true
echo "[8] all five BASELINE configurations"
timeout -k 10 600 python3 tools/config_report.py > $out/${tag}_configs.txt 2>&1; grep -v amdgpu $out/${tag}_configs.txt
echo collected
