export TMPDIR=/tmp
for M in ${FIT_MS:-256 512 1024 2048}; do
  rm -rf gpurun_out/fit_$M
  KR_TRAIN_M=$M timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fit_$M -- python3 tools/train_only.py cfg3 > gpurun_out/fit_$M.log 2>&1 || exit 1
  f=$(find gpurun_out/fit_$M -name "*kernel_stats.csv" | head -1)
  echo "M=$M"; python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"]
    if any(k in n for k in ("mlp_", "tail")):
        print(f"  {n[:44]:44s} avg {float(row['AverageNs'])/1e3:8.1f} us")
PY
done
