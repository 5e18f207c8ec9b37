#!/bin/bash
# Dev tool (GPU box): headline A/B of two builds of libknode_rod.so in alternating processes.
#   [ROUNDS=2] tools/ab_headline.sh <tag> <lib> [<lib> ...]     (paths relative to the repository root)
# Prints, per build and round, rod-steps/s over 1000-step launches and as the median of five 20-step chunks (the
# driver's command), and writes everything to gpurun_out/<tag>_ab.txt.
set -u
tag=$1; shift; rounds=${ROUNDS:-2}
out=gpurun_out/${tag}_ab.txt
mkdir -p gpurun_out; : > "$out"
for r in $(seq 1 "$rounds"); do
  for lib in "$@"; do
    for steps in 1000 20; do
      if [ "$steps" = 1000 ]; then extra="--warmup 60 --chunks 2"; else extra="--warmup 5"; fi
      line=$(KR_LIB_PATH=$PWD/$lib timeout -k 10 300 python3 bench.py --steps $steps $extra --no-cpu --no-extra 2>/dev/null | tail -1)
      echo "$line" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('round $r  %-52s steps %4d: %.3f M rod-steps/s  (%.3f ms/step, unconverged %s)' % ('$lib', d['steps'], d['value']/1e6, d['ms_per_step'], d.get('config',{}).get('unconverged_rod_steps', d.get('unconverged_rod_steps'))))" | tee -a "$out"
    done
  done
done
