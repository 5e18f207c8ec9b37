#!/bin/bash
# GPU box: SQ / memory counters of the training kernels (cfg3).   bash tools/pmc_train.sh <tag>
tag=${1:-pt}
export TMPDIR=/tmp
out=gpurun_out
run() {  # name, counters...
  local name=$1; shift
  rm -rf $out/${tag}_$name
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/${tag}_$name -- python3 tools/train_only.py cfg3 > /dev/null 2>&1 || echo "$name pass failed"
  f=$(find $out/${tag}_$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"][:40]
    if "mlp_" not in k: continue
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
    n[(k, row["Counter_Name"])] += 1
for k in acc:
    print(k, {c: round(v / n[(k, c)], 1) for c, v in acc[k].items()})
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS
run sq2 SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE
