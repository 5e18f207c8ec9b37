#!/bin/bash
# GPU box: kernel-level timing of the fused training epoch (cfg3: 28-64-64-25, Q = 193 536; cfg4 shard: 28-512-25, Q = 59 392)
#   usage: bash tools/prof_train.sh <tag>      -> gpurun_out/<tag>_train_{cfg3,cfg4}_stats.csv
tag=${1:-tr}
export TMPDIR=/tmp
for cfg in cfg3 cfg4; do
  rm -rf gpurun_out/${tag}_train_${cfg}
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_train_${cfg} -- python3 tools/train_only.py $cfg > gpurun_out/${tag}_train_${cfg}.log 2>&1 || exit 1
  f=$(find gpurun_out/${tag}_train_${cfg} -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/${tag}_train_${cfg}_stats.csv
  echo "== $cfg"; python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if any(k in n for k in ("mlp_", "loss", "adam", "pack_", "reduce_", "tail")):
        print(f"{n[:60]:60s} calls {row['Calls']:>5s} avg {float(row['AverageNs'])/1e3:8.1f} us")
PY
done
