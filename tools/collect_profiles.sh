#!/bin/bash
# GPU box: bench line + rocprofv3 kernel stats + HBM counters (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes).  Raw output -> gpurun_out/<tag>_*; tools/summarise_profiles.py
# turns it into the files committed under profiles/.
#   usage: bash tools/collect_profiles.sh r01c
set -e
tag=${1:-prof}
out=gpurun_out
export TMPDIR=/tmp
mkdir -p $out
timeout -k 10 400 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -1 $out/${tag}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --no-cpu > $out/${tag}_stats.json 2> $out/${tag}_stats.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 bench.py --no-cpu > /dev/null 2> $out/${tag}_fetch.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 bench.py --no-cpu > /dev/null 2> $out/${tag}_write.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $out/${tag}_sq -- python3 bench.py --no-cpu > /dev/null 2> $out/${tag}_sq.err || echo "SQ pass failed (non-fatal)"
echo collected
